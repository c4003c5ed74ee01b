// trg_kernels.h -- host-visible launch interface of the HIP kernels (internal to libtoyraygun_hip.so).
// Two implementations are linked in: *_fast (shipped) and *_strict (-ffp-contract=off parity build).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/trg.h"

#ifndef TRG_WIDE8
#define TRG_WIDE8 0   // 1 (round-5 experiment; scripts/exp_build.sh <name> -DTRG_WIDE8=1): scenes in HBM are traversed through the 80-byte COMPRESSED 8-WIDE
                      // nodes of q8node.h (host builder only) instead of the quantised 4-wide nodes
#endif
#ifndef TRG_EXPERIMENTS
#define TRG_EXPERIMENTS 0   // 1: experiments/lib/libtoyraygun_hip_exp.so -- this library + the schedules kept in experiments/ (pool, wavefront)
#endif

namespace trg {

#ifndef TRG_BLOCK
#define TRG_BLOCK 256
#endif
constexpr int kBlock = TRG_BLOCK;  // threads per workgroup: 1, 2 or 4 wavefronts of 64
constexpr int kTileW = kBlock >= 128 ? 16 : 8;   // a workgroup renders a kTileW x kTileH pixel tile,
constexpr int kTileH = kBlock >= 256 ? 16 : 8;   // one 8x8 sub-tile per wavefront
constexpr int kWaves = kBlock / 64;
static_assert(kBlock == 64 || kBlock == 128 || kBlock == 256, "workgroup must be 1, 2 or 4 wavefronts");
constexpr int kCounterSlots = 64;  // ray counters are spread over 64 slots to avoid same-address atomics
constexpr int kCounterWords = 8;   // primary, bounce, shadow, shaded, node_fetches, tri_tests, 2 spare
#ifndef TRG_MAX_LDS_SCENE_KB
#define TRG_MAX_LDS_SCENE_KB 40
#endif
constexpr uint32_t kMaxLdsScene = TRG_MAX_LDS_SCENE_KB * 1024u;  // scenes up to this size are staged in LDS per workgroup
// render_kernel on an HBM-resident scene: park throughput + radiance in LDS while a ray pair is traced (6 floats per thread next
// to the 3 of the running average), and keep this many traversal-stack levels in LDS (deeper ones go to global scratch)
#ifndef TRG_PARK_PATH
#define TRG_PARK_PATH 1
#endif
#ifndef TRG_PARK_OFFSET
#define TRG_PARK_OFFSET 1   // ... and the pixel's Halton offset behind them (one more word per thread)
#endif
#ifndef TRG_STACK_LDS_LEVELS
#define TRG_STACK_LDS_LEVELS 12
#endif

// The scene lives in ONE device allocation.  A scene small enough for LDS starts with the part a workgroup stages:
// [nodes | tris | normals | colors | mats | Halton tables], every section 16-byte aligned, so that it goes into LDS with a single
// stream of 16-byte copies.  Every scene then has what the HBM kernels traverse: [quantised 4-wide nodes | 128-byte leaf records].
struct SceneDesc {
    const unsigned char *blob;
    uint32_t off_nodes, off_tris, off_normals, off_colors, off_mats, blob_bytes;
    uint32_t n_nodes, n_tris;
    uint32_t off_nodes4, n_nodes4;  // 4-wide nodes (HBM traversal); they sit after the LDS-staged part of the blob
    uint32_t lds_stage_bytes;       // bytes a workgroup stages into LDS (everything before the 4-wide nodes)
    uint32_t off_htab;              // Halton group tables (kHtabFloats floats), inside the staged region
    uint32_t off_fat, n_fat;        // leaf records of the HBM traversal: geometry + attributes, 128 bytes each, 128-byte aligned, leaf order
    // LDS-resident scenes, shipped build (TRG_TRI_PLANES, round 4): the triangle test on three precomputed PLANES per triangle -- 48 bytes in
    // leaf order like the Moeller-Trumbore records (off_tris), kept outside the staged part and copied over them at staging time --
    // and, inside the staged part, one u16 per record: (original index << 2) | (material id & 3), the test's mask and the hit's primitive
    uint32_t off_tris_alt, off_meta, n_tris_rec;   // n_tris_rec = 48-byte records of the staged part (leaf order)
    // scenes traversed from HBM, shipped build (TRG_TRI_PLANES_HBM): a second set of 128-byte leaf records whose rows 0..2 are the triangle's
    // three planes, then the original index and the material id (floats 12, 13: TRG_REC_META_FIRST), then the attributes (floats 14..31)
    uint32_t off_fat_planes;
    // FLAT primitive list (round-5 EXPERIMENT, experiments library with TRG_FLAT_PRIMS=1 only; scenes of at most kFlatMaxPrims primitives after the quads are paired -- the Cornell box has 18):
    // n_flat x 64 bytes in the blob, outside the staged part -- rows 0..2 the plane record of the triangle or of the quad's parallelogram (as
    // off_tris_alt), row 3 = (original index of the triangle / of the quad's X, original index of the quad's Y or ~0, material id = mask, 1 = quad).
    // Such a scene is not walked through its tree at all: every ray tests every primitive in record order (trg_device.h traverse_flat).
    uint32_t off_flat, n_flat;
    // BOX leaves of the shipped build's HBM traversal (TRG_BOX_LEAVES_HBM, round 5): a second array of quantised 4-wide nodes in which a box's subtree is
    // ONE leaf -- child code ~(((n_fat + box index) << 3) | 6): a box is addressed like a leaf record behind the plane records, same 128-byte stride, so the
    // traversal's address arithmetic does not know about boxes -- and 64 bytes per box: rows 0..2 = (a_k, d_k), l_k = a_k . (P - center) + d_k the box's own
    // frame, row 3 = (its first leaf record, material id, face table low, high: 7 bits per face f = 2 k + (l_k > 0), 0..3 in the low word, 4..5 in
    // the high one -- X-record offset 0..10 | swap << 4 | negate s << 5 | negate t << 6; bit 31 of the high word: a lone quad dressed as a box).  A scene without boxes: off_nodes4_box = off_nodes4, n_boxrec = 0.
    uint32_t off_nodes4_box, off_boxrec, n_boxrec;
    // the planes are stored relative to this point (the centre of the scene's bounding box) and a ray's origin is shifted by it when its
    // traversal begins: n . o - d0 then cancels numbers of the size of the scene instead of its distance from the coordinate origin
    float center[3];
};
constexpr uint32_t kFatRecBytes = 128u;
#ifndef TRG_FLAT_MAX_PRIMS
#define TRG_FLAT_MAX_PRIMS 32
#endif
constexpr uint32_t kFlatMaxPrims = TRG_FLAT_MAX_PRIMS, kFlatPrimBytes = 64u;   // 0: never a flat list

// albedo textures (trg_load_textures): per-corner texture coordinates, per-triangle texture id (0 = none), a table of
// (first texel, width, height, -) per texture and the RGBA8 texels of all textures back to back.  uv == nullptr: no textures.
struct TexDesc {
    const float *uv;
    const uint32_t *ids;
    const uint32_t *table;
    const uint32_t *texels;
};

// traversal stack: the first `klds` levels live in LDS, deeper ones in a global scratch column per thread
struct StackDesc {
    int *overflow;      // (levels - klds) x grid_threads ints, or nullptr
    uint32_t klds;
};

struct RenderParams {
    trg_uniforms u;
    SceneDesc sc;
    const uint32_t *offsets;       // width*height Halton index offsets
    float *accum;                  // width*height float4, row 0 = scene bottom
    unsigned long long *counters;  // kCounterSlots * kCounterWords
    uint32_t frame_begin, spp, bounces, row0, rows, tiles_x;
    uint32_t stack_off;            // byte offset of the traversal stacks in dynamic LDS
    uint32_t red_off;              // byte offset of the counter-reduction scratch in dynamic LDS
    uint32_t pool_off;             // render_fp_kernel: byte offset of the parked per-frame radiances (experiments: render_pool_kernel's path pool)
    TexDesc tex;
    uint32_t acc_off;              // render_kernel on an HBM-resident scene: byte offset of the parked running average (3 x kBlock floats)
    // tail compaction (trg_tail.inc.h): bounces >= tail_k of the frames of a chunk run in a second launch on compacted paths
    uint32_t tail_k, tail_k_end, tail_cap, tail_band_pixels;   // this launch runs bounces [tail_k, tail_k_end) (the head: [0, tail_k))
    void *tail_queue;              // float4[3] x tail_cap entries per head wavefront: the queue this launch reads (tail) / writes (head)
    uint32_t *tail_count;          // entries per head wavefront
    void *tail_queue_out;          // a tail launch that stops before the last bounce compacts its survivors into this queue
    uint32_t *tail_count_out;
    void *tail_radbuf;             // float4 [frames of the chunk][pixels of the band]
    uint32_t fsplit, fp_rounds;    // render_fp_kernel: frame lanes per workgroup (2 or 4), rounds parked per fold
    StackDesc stack;
    // workgroup -> tile (trg_kernels.hip block_tile): tiles_x x tiles_y tiles of the launch; xcd_cols = 0: image columns from the centre
    // outwards; 1, 2, 4, 8: the workgroups of one XCD (blockIdx.x % 8) own one of xcd_cols x (8 / xcd_cols) contiguous screen regions
    uint32_t tiles_y, xcd_cols;
    // INTERLEAVED row bands (trg_render_bands; SURVEY 8e's micro-bands, 8 rows each): il_n > 1 = this launch renders the micro-bands
    // il_r, il_r + il_n, il_r + 2 il_n ... of the image and stores them COMPACTLY: local row l (micro-band l / 8 of this rank) is image
    // row ((l / 8) * il_n + il_r) * 8 + l % 8 for the camera ray and the pixel's Halton offset, and row row0 + l of the accumulation
    // buffer.  il_n <= 1: rows [row0, row0 + rows) of the image, stored where they are.
    uint32_t il_n, il_r;
    // PERSISTENT workgroups with one job queue per XCD (TRG_OPT_TILE_ORDER 66; render_regen_kernel only, round 4): xq != nullptr = the launch
    // has about as many workgroups as the chip holds; each one reads the XCD it runs on (HW_REG_XCC_ID) and pops jobs -- (tile, frame lane)
    // pairs, xq_jobs per queue -- from THAT XCD's queue, whose tiles are one of the 2 x 4 screen regions of the XCD-aware order (xcd_cols), and
    // steals from the next queues when its own is empty.  An L2 then serves one screen region, whichever XCD the dispatcher's round robin
    // hands a workgroup to, and no XCD idles behind a slower one.  xq: 8 counters, zeroed before the launch.
    uint32_t *xq;
    uint32_t xq_jobs;
};
constexpr uint32_t kMicroBandRows = 8;   // = the rows of a wavefront's 8x8 sub-tile: a wavefront never straddles two micro-bands
constexpr uint32_t kXcdPersist = 64;   // tile order 64 + n: persistent workgroups popping from per-XCD queues over the n x (8 / n) regions (regeneration kernel)
constexpr uint32_t kXcdBlocks = 32;    // tile order 32: 8 x 8-tile super-blocks, one 2 x 4-tile sub-block per XCD
constexpr uint32_t kXcdStripes = 16;   // tile orders 17, 18, 20, 24: stripes of 1, 2, 4, 8 tile rows dealt round robin over the XCDs
constexpr uint32_t kXcds = 8;   // XCDs of an MI355X: workgroups are dealt round-robin over them (MI355X_MICROARCH.md, workgroup dispatch)

// Workgroup slot -> tile of a launch over tiles_x x tiles_y tiles (any bijection onto the tiles is correct; this is about time only).
//  * xcd_cols == 0: image columns from the centre outwards (rows inner).  It starts the tiles a camera usually points at first, so the
//    tail of the launch is made of the cheap edge tiles (C2: the 16:9 side bars) instead of leaving CUs idle behind a few expensive
//    ones (+9 % on C2).
//  * xcd_cols = 1, 2, 4, 8: XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MiB L2, so the workgroups
//    with equal slot % 8 share an L2.  They get ONE of xcd_cols x (8 / xcd_cols) contiguous regions of the screen (column strips x row
//    bands) and walk it column by column away from the image centre: an L2 then sees the rays of one screen region -- one part of the
//    scene -- instead of every eighth tile of the whole picture.  Regions differ in size by a row or a column at most, so the launch
//    is padded to 8 x the largest region and the few slots beyond a region's end return false.  All regions are at the same distance
//    from the centre at the same time.  (Measured on C4: a third less memory-side traffic and 14 % MORE time -- trg_capi.cpp
//    choose_xcd_cols -- so the column order is the default.)
// Host and device (the CPU tests check the bijection without a GPU through trg_debug_tile_of_slot).
#if defined(__HIPCC__)
#define TRG_HD __host__ __device__
#else
#define TRG_HD
#endif
TRG_HD inline bool tile_of_slot(uint32_t tiles_x, uint32_t tiles_y, uint32_t xcd_cols, uint32_t slot, uint32_t &bx, uint32_t &by) {
    const uint32_t cleft = (tiles_x - 1u) / 2u;
    if (xcd_cols == 0u) {
        const uint32_t crank = slot / tiles_y;
        by = slot % tiles_y;
        bx = (crank & 1u) ? cleft + 1u + crank / 2u : cleft - crank / 2u;
        return crank < tiles_x;
    }
    const uint32_t x = slot % kXcds, j = slot / kXcds;
    if (xcd_cols == kXcdBlocks) {
        // BLOCKS (round 4): the picture in super-blocks of 8 x 8 tiles, walked in columns from the centre outwards; the 8 XCDs work on ONE
        // super-block at a time, each on a compact 2 x 4-tile sub-block of it (which XCD takes which sub-block rotates from super-block to
        // super-block, so that the partial super-blocks at the picture's edge cost every XCD the same).  Neighbouring regions cost alike
        // and the dispatcher's strict round robin never waits long, while an L2 serves 8 adjacent tiles instead of 8 scattered ones.
        const uint32_t nsy = (tiles_y + 7u) / 8u, nsx = (tiles_x + 7u) / 8u;
        const uint32_t sb = j / 8u, t = j % 8u;
        const uint32_t sbc = sb / nsy, sby = sb - sbc * nsy;
        if (sbc >= nsx) return false;
        const uint32_t scl = (nsx - 1u) / 2u;
        const uint32_t sbx = (sbc & 1u) ? scl + 1u + sbc / 2u : scl - sbc / 2u;
        const uint32_t pos = (x + sb) % kXcds;                                      // this XCD's sub-block of the super-block: 4 across, 2 down
        bx = sbx * 8u + (pos % 4u) * 2u + t % 2u;
        by = sby * 8u + (pos / 4u) * 4u + t / 2u;
        return bx < tiles_x && by < tiles_y;
    }
    if (xcd_cols >= kXcdStripes) {
        // STRIPES (round 4): XCD x owns the tile rows whose group of S rows has number x mod 8 -- S = xcd_cols - 16 in {1, 2, 4, 8} -- i.e.
        // horizontal stripes of S tile rows dealt round robin over the XCDs: every XCD gets an eighth of every part of the picture (the
        // regions above differ in cost and the dispatcher deals the XCDs their workgroups strictly in turn), and the workgroups an L2 serves
        // at one time still sit side by side in a few stripes.  Columns from the centre outwards, the XCD's rows inner.
        const uint32_t S = xcd_cols - kXcdStripes;
        const uint32_t hmax = ((tiles_y + kXcds * S - 1u) / (kXcds * S)) * S;   // rows of the XCD with the most
        const uint32_t c = j / hmax, k = j - c * hmax;
        const uint32_t row = ((k / S) * kXcds + x) * S + k % S;
        if (row >= tiles_y || c >= tiles_x) return false;
        bx = (c & 1u) ? cleft + 1u + c / 2u : cleft - c / 2u;
        by = row;
        return true;
    }
    const uint32_t rc = xcd_cols, rq = kXcds / rc;
    const uint32_t ci = x % rc, qi = x / rc;
    const uint32_t c0 = ci * tiles_x / rc, c1 = (ci + 1u) * tiles_x / rc;
    const uint32_t r0 = qi * tiles_y / rq, r1 = (qi + 1u) * tiles_y / rq;
    const uint32_t W = c1 - c0, H = r1 - r0;
    if (H == 0u) return false;
    const uint32_t c = j / H, r = j - c * H;
    if (c >= W) return false;
    if (rc == 1u) bx = (c & 1u) ? cleft + 1u + c / 2u : cleft - c / 2u;   // one strip per row band: columns from the centre outwards, as above
    else bx = (2u * ci < rc) ? c1 - 1u - c : c0 + c;                      // strips left of the centre walk to the left edge, the others to the right edge
    by = r0 + r;
    return true;
}

struct TraceParams {
    SceneDesc sc;
    const trg_ray *rays;
    void *out;  // trg_isect[n] or float[n]
    uint32_t n;
    uint32_t stack_off;
    StackDesc stack;
};

// Radical-inverse tables for the five smallest odd bases (Halton dimensions 1..5: the pixel y offset and the four
// samples of bounce 0), staged in LDS with the scene: T_b[r] = radical inverse of r read as a K-digit base-b number.
// The shipped megakernel on an LDS-resident scene takes K digits per lookup instead of one digit per five VALU
// instructions (trg_device.h halton_c); everything else (strict build, HBM scenes, other dimensions) keeps the digits.
// The shipped build's triangle test on an LDS-resident scene: 1 = three planes per triangle (17 arithmetic instructions), 0 = Moeller-Trumbore
// on (v0, e1, e2) like the strict build (31).  trg_device.h tri_test_planes.
#ifndef TRG_TRI_PLANES
#define TRG_TRI_PLANES 1
#endif
#ifndef TRG_REC_META_FIRST
#define TRG_REC_META_FIRST 1   // plane-form leaf records: 1 = the original index and the material id are floats 12, 13 -- everything a triangle TEST reads
                               // sits in the first 64 bytes of the record, the attributes follow at floats 14..31; 0 = attributes at 12..29, index / mask last
#endif
#ifndef TRG_TRI_PLANES_HBM
#define TRG_TRI_PLANES_HBM 1   // the same for scenes traversed from HBM (their own set of leaf records, SceneDesc::off_fat_planes)
#endif
#ifndef TRG_BOX_LEAVES
#define TRG_BOX_LEAVES 1       // scenes staged in LDS: addCube's twelve triangles become ONE leaf (bvh_build.h kLeafBox): twelve triangle tests in the strict
                               // build, one slab test in the parallelepiped's own frame in the shipped one (trg_device.h trav_box_planes)
#endif
#ifndef TRG_BOX_LEAVES_HBM
#define TRG_BOX_LEAVES_HBM 1   // the same box leaf for scenes traversed from HBM (shipped build: SceneDesc::off_nodes4_box / off_boxrec; the strict build keeps the plain tree)
#endif
#ifndef TRG_HALTON_TABLES
#define TRG_HALTON_TABLES 1
#endif
struct HtabSpec { uint32_t base, digits, radix, offset; };
constexpr HtabSpec kHtab[5] = { { 3, 4, 81, 0 }, { 5, 3, 125, 81 }, { 7, 2, 49, 206 }, { 11, 2, 121, 255 }, { 13, 2, 169, 376 } };
constexpr uint32_t kHtabFloats = 545, kHtabBytes = 2192;  // 545 * 4 rounded up to 16

// Node layouts.  Scenes traversed from HBM: the quantised 4-wide tree (q4node.h, 64 bytes per node) and 128-byte leaf records.  Scenes staged in
// LDS: the SIGN-ORDERED BVH2 node (trg_device.h trav_node_step_signed): 192 bytes used + 16 of padding.  (A plain 64-byte BVH2 node, a
// sign-ordered 4-wide float node, an octant-threaded stackless list and a half-precision node were built and measured slower: NOTEBOOK.md.)
constexpr uint32_t kLdsNodeBytes = 208u;

#define TRG_DECL_LAUNCHERS(SFX)                                                                                   \
    hipError_t launch_render_##SFX(const RenderParams &p, bool lds_scene, bool counters, uint32_t grid,          \
                                   size_t lds_bytes, hipStream_t s);                                             \
    hipError_t launch_render_regen_##SFX(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s); \
    hipError_t launch_regen_accumulate_##SFX(const RenderParams &p, uint32_t grid, hipStream_t s);               \
    hipError_t launch_render_fp_##SFX(const RenderParams &p, bool lds_scene, bool counters, uint32_t grid,       \
                                      size_t lds_bytes, hipStream_t s);                                          \
    hipError_t launch_render_head_##SFX(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s); \
    hipError_t launch_render_tail_##SFX(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s); \
    hipError_t launch_tail_accumulate_##SFX(const RenderParams &p, hipStream_t s);                               \
    hipError_t launch_tail_sort_##SFX(const RenderParams &p, uint32_t grid, uint32_t mode, const float *lo3, const float *inv3, hipStream_t s); \
    hipError_t launch_trace_##SFX(const TraceParams &p, bool lds_scene, bool any_hit, size_t lds_bytes,          \
                                  hipStream_t s);                                                                \
    hipError_t launch_halton_##SFX(const uint32_t *i, const uint32_t *d, uint32_t n, float *out, hipStream_t s); \
    hipError_t launch_halton_tab_##SFX(const float *htab, const uint32_t *i, const uint32_t *d, uint32_t n, float *out,   \
                                       hipStream_t s);                                                           \
    hipError_t launch_raygen_##SFX(const trg_uniforms &u, const uint32_t *offsets, trg_ray *out, hipStream_t s); \
    hipError_t launch_sample_##SFX(const trg_uniforms &u, const float *p3, const float *n3, const float *r4,     \
                                   uint32_t n, float *out12, hipStream_t s);                                     \
    hipError_t launch_postprocess_##SFX(const float *accum, uint32_t w, uint32_t h, uint8_t *rgba8, int flip_y,  \
                                        hipStream_t s);                                                          \
    hipError_t launch_offsets_##SFX(uint32_t seed, uint32_t n, uint32_t *out, hipStream_t s);                   \
    hipError_t launch_unpack_bands_##SFX(const float *compact, float *image, uint32_t w, uint32_t h, uint32_t n, uint32_t B, \
                                         hipStream_t s);                                                        \
    hipError_t launch_xcc_probe_##SFX(uint32_t n_blocks, uint32_t *out, hipStream_t s);

TRG_DECL_LAUNCHERS(fast)
TRG_DECL_LAUNCHERS(strict)

}  // namespace trg

#if TRG_EXPERIMENTS
#include "../../experiments/trg_exp.h"
#endif
