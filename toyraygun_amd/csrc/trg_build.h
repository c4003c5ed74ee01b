// trg_build.h -- device-side acceleration-structure build (trg_build.hip), internal to libtoyraygun_hip.so.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace trg {

hipError_t gpu_build_lbvh(const float *d_pos, const uint32_t *d_idx, const uint32_t *d_masks, uint32_t ntris,
                          const float scene_lo[3], const float scene_hi[3], float pad, float4 *d_nodes4, float4 *d_tris,
                          uint32_t *n_nodes4, uint32_t *depth4, hipStream_t s, int mode,   // mode: 0 = Karras LBVH hierarchy, 1 = binned SAH by levels, 2 = PLOC merges
                          // quads (bvh_build.h pair_quads): the builders' primitives -- px[i] a triangle or the X of a quad, py[i] its Y or ~0u --,
                          // nprims of them (>= 2); d_quad_rec: ntris bytes, zeroed, receives a 1 at every X record.  d_px == nullptr: no pairing
                          const uint32_t *d_px = nullptr, const uint32_t *d_py = nullptr, uint32_t nprims = 0, unsigned char *d_quad_rec = nullptr);

// float 4-wide nodes (8 float4 each) -> quantised 64-byte nodes (q4node.h); d_out holds n_nodes4 * 64 bytes
hipError_t gpu_quantize_nodes4(const float4 *d_nodes4, uint32_t n_nodes4, void *d_out, hipStream_t s);

// 48-byte geometry records (leaf order) + normals / colours in original order (9 floats per triangle) -> 128-byte leaf records
// (trg_device.h kRecV4) at d_out
// planes: the shipped build's form -- rows 0..2 the triangle's three planes (computed in double, as trg_capi.cpp fill_plane_record does on the
// host), then the original index and the material id (floats 12, 13), then the attributes (floats 14..31): TRG_REC_META_FIRST
// d_quad_rec (planes only): per record, 1 = the X of a quad leaf -- its planes are the parallelogram's (X.e1, the next record's e2)
hipError_t gpu_fatten_records(const float4 *d_tris48, const float *d_normals, const float *d_colors, uint32_t ntris, void *d_out, bool planes, const float center[3], hipStream_t s,
                              const unsigned char *d_quad_rec = nullptr);

}  // namespace trg
