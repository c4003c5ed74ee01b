// trg_kernels.hip -- the gfx950 path-tracing megakernel and its stage-level test kernels.
//
// One launch of render_kernel does what the reference does with 14 dispatches per frame x spp frames
// (src/engine/Metal/MetalRenderer.mm:400-515): raygen -> bounces x [nearest-hit -> primaryHit ->
// any-hit -> shadowHit] -> accumulate, for every frame of the requested range, with the ray, the
// throughput and the running average held in registers.  HBM traffic per pixel per launch: one
// 4-byte Halton offset read and one 16-byte float4 write (+16 B read when continuing an average).
//
// Built twice: TRG_STRICT=0 -> *_fast symbols, TRG_STRICT=1 (-ffp-contract=off) -> *_strict symbols.
#include "trg_device.h"
#include "trg_kernels.h"

using namespace trgdev;

#if TRG_STRICT
#define SFX(name) name##_strict
#else
#define SFX(name) name##_fast
#endif

namespace SFX(trgk) {

// Stage the scene blob into LDS (16-byte copies by the whole workgroup) or point at it in HBM.
template <bool LDS_SCENE>
TRG_DEV SceneView scene_view(const trg::SceneDesc &sc, unsigned char *smem) {
    SceneView v;
    if (LDS_SCENE) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.blob);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = sc.blob_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += trg::kBlock) dst[i] = src[i];
        __syncthreads();
        v.nodes = reinterpret_cast<const v4f *>(smem + sc.off_nodes);
        v.tris = reinterpret_cast<const v4f *>(smem + sc.off_tris);
        v.normals = reinterpret_cast<const float *>(smem + sc.off_normals);
        v.colors = reinterpret_cast<const float *>(smem + sc.off_colors);
        v.mats = reinterpret_cast<const uint32_t *>(smem + sc.off_mats);
    } else {
        v.nodes = reinterpret_cast<const v4f *>(sc.blob + sc.off_nodes);
        v.tris = reinterpret_cast<const v4f *>(sc.blob + sc.off_tris);
        v.normals = reinterpret_cast<const float *>(sc.blob + sc.off_normals);
        v.colors = reinterpret_cast<const float *>(sc.blob + sc.off_colors);
        v.mats = reinterpret_cast<const uint32_t *>(sc.blob + sc.off_mats);
    }
    return v;
}

TRG_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool LDS_SCENE, bool COUNT>
__global__ __launch_bounds__(trg::kBlock) void render_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SceneView sc = scene_view<LDS_SCENE>(p.sc, smem);
    LdsStack<trg::kBlock> stk;
    stk.base = reinterpret_cast<int *>(smem + p.stack_off) + threadIdx.x;

    // workgroup -> 16x16 tile, wavefront -> 8x8 sub-tile, lane -> pixel (8 consecutive pixels of a row
    // per 8 lanes: each wavefront writes eight 128-byte row segments)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t bx = blockIdx.x % p.tiles_x, by = blockIdx.x / p.tiles_x;
    const uint32_t x = bx * trg::kTile + (wave & 1) * 8 + (lane & 7);
    const uint32_t y = p.row0 + by * trg::kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = (x < p.u.width) && (y < p.row0 + p.rows) && (y < p.u.height);
    const uint32_t pix = y * p.u.width + x;

    uint32_t n_primary = 0, n_bounce = 0, n_shadow = 0, n_shaded = 0;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0;

    const uint32_t offset = valid ? p.offsets[pix] : 0u;
    v4f *accum = reinterpret_cast<v4f *>(p.accum);
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    if (valid && p.frame_begin > 0) { const v4f a = accum[pix]; acc = mk(a.x, a.y, a.z); }
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);

    for (uint32_t f = p.frame_begin; f < p.frame_begin + p.spp; ++f) {
        const uint32_t hidx = offset + f;  // Raytracing.metal:67: offset + uniforms.frameIndex (wraps mod 2^32)
        V3 o, d;
        raygen(p.u, x, y, hidx, o, d);
        V3 thr = mk(1.0f, 1.0f, 1.0f);  // ray.color
        V3 rad = mk(0.0f, 0.0f, 0.0f);  // the render target texel of this frame
        uint32_t rmask = 3u;            // RAY_MASK_PRIMARY
        bool active = valid;
        if (active) n_primary++;

        for (uint32_t b = 0; b < p.bounces; ++b) {
            if (__ballot(active) == 0ull) break;  // whole wavefront terminated
            if (active) {
                if (b > 0) n_bounce++;
                Hit h;
                const bool found = traverse<false, COUNT, trg::kBlock>(sc, o, d, INFINITY, rmask, h, stk, cnt);
                if (!found) {
                    active = false;  // Raytracing.metal:139-144
                } else {
                    const uint32_t mat = sc.mats[h.prim];
                    if (mat == TRG_MATERIAL_DEFAULT) {
                        n_shaded++;
                        // Raytracing.metal:150-199
                        const V3 P = o + d * h.t;
                        const float cx = 1.0f - h.u - h.v, cy = h.u;  // weights of vertex 0, 1
                        const V3 vcol = interp_attr(sc.colors, h.prim, cx, cy);
                        const V3 nrm = normalize(interp_attr(sc.normals, h.prim, cx, cy));
                        float r[4];
                        // opaque copy: stops LICM from hoisting every bounce's Halton digits (all 60
                        // dimensions) out of the bounce loop and keeping them live in VGPRs
                        uint32_t hi = hidx;
                        asm volatile("" : "+v"(hi));
                        halton4(hi, b, r);
                        const LightSample ls = sample_area_light(p.u, r[0], r[1], P, nrm);
                        thr = thr * vcol;
                        const V3 so = P + nrm * 1e-3f;
                        const float smax = ls.dist - 1e-3f;
                        const V3 scol = ls.color * thr;
                        if (smax >= 0.0f) {  // inactive shadow rays are not traced (MPS skips maxDistance < 0)
                            n_shadow++;
                            Hit sh;
                            const bool occluded = traverse<true, COUNT, trg::kBlock>(sc, so, ls.dir, smax, 1u, sh, stk, cnt);
                            if (!occluded) rad = rad + scol;  // Raytracing.metal:240-241
                        }
                        const V3 sd = sample_cosine_hemisphere(r[2], r[3]);
                        d = align_hemisphere(sd, nrm);
                        o = so;
                        rmask = 1u;  // RAY_MASK_SECONDARY
                    } else if (mat == TRG_MATERIAL_EMISSIVE) {
                        rad = light_color;  // Raytracing.metal:200-209: overwrites the texel, ends the path
                        active = false;
                    } else {
                        rad = mk(1.0f, 0.0f, 1.0f);  // Raytracing.metal:210-214: magenta, ray left as is
                    }
                }
            }
        }
        // Accumulate.metal:19-39
        if (f == 0) {
            acc = rad;
        } else {
            const V3 prev = acc * (float)f;
            const V3 c = rad + prev;
            const float f1 = (float)(f + 1u);
            acc = mk(c.x / f1, c.y / f1, c.z / f1);
        }
    }
    if (valid) {
        v4f outv; outv.x = acc.x; outv.y = acc.y; outv.z = acc.z; outv.w = 1.0f;
        accum[pix] = outv;
    }

    // ray counters: wavefront reduce -> workgroup reduce in LDS -> one atomic per counter per workgroup,
    // spread over kCounterSlots slots
    uint32_t vals[6] = { n_primary, n_bounce, n_shadow, n_shaded, cnt.nodes, cnt.tris };
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + p.red_off);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (!COUNT && k >= 4) break;
        const uint32_t s = wave_sum(vals[k]);
        if (lane == 0) red[wave * 8 + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < (COUNT ? 6 : 4)) {
        const uint32_t k = threadIdx.x;
        const unsigned long long s = (unsigned long long)red[k] + red[8 + k] + red[16 + k] + red[24 + k];
        if (s) atomicAdd(&p.counters[(blockIdx.x % trg::kCounterSlots) * trg::kCounterWords + k], s);
    }
}

// ---- a7 / a12 in isolation: one ray per thread ----
template <bool LDS_SCENE, bool ANY>
__global__ __launch_bounds__(trg::kBlock) void trace_kernel(const trg::TraceParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SceneView sc = scene_view<LDS_SCENE>(p.sc, smem);
    LdsStack<trg::kBlock> stk;
    stk.base = reinterpret_cast<int *>(smem + p.stack_off) + threadIdx.x;
    const uint32_t i = blockIdx.x * trg::kBlock + threadIdx.x;
    if (i >= p.n) return;
    const trg_ray r = p.rays[i];
    const V3 o = mk(r.origin[0], r.origin[1], r.origin[2]), d = mk(r.direction[0], r.direction[1], r.direction[2]);
    Counters cnt; cnt.nodes = 0; cnt.tris = 0;
    Hit h; h.t = -1.0f; h.prim = -1; h.u = 0.0f; h.v = 0.0f;
    bool found = false;
    if (r.maxDistance >= 0.0f) found = traverse<ANY, false, trg::kBlock>(sc, o, d, r.maxDistance, r.mask, h, stk, cnt);
    if (ANY) {
        reinterpret_cast<float *>(p.out)[i] = found ? h.t : -1.0f;
    } else {
        trg_isect is;
        is.distance = found ? h.t : -1.0f;
        is.primitiveIndex = found ? h.prim : -1;
        is.coordinates[0] = found ? (1.0f - h.u - h.v) : 0.0f;
        is.coordinates[1] = found ? h.u : 0.0f;
        reinterpret_cast<trg_isect *>(p.out)[i] = is;
    }
}

__global__ void halton_kernel(const uint32_t *i, const uint32_t *d, uint32_t n, float *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = halton_any(i[k], d[k]);
}

__global__ void raygen_kernel(const trg_uniforms u, const uint32_t *offsets, trg_ray *out) {
    const uint32_t x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= u.width || y >= u.height) return;
    const uint32_t pix = y * u.width + x;
    V3 o, d;
    raygen(u, x, y, offsets[pix] + u.frameIndex, o, d);
    trg_ray r;
    r.origin[0] = o.x; r.origin[1] = o.y; r.origin[2] = o.z;
    r.mask = 3u;
    r.direction[0] = d.x; r.direction[1] = d.y; r.direction[2] = d.z;
    r.maxDistance = INFINITY;
    r.color[0] = 1.0f; r.color[1] = 1.0f; r.color[2] = 1.0f; r.color[3] = 0.0f;
    out[pix] = r;
}

__global__ void sample_kernel(const trg_uniforms u, const float *p3, const float *n3, const float *r4, uint32_t n, float *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const V3 P = mk(p3[k * 3], p3[k * 3 + 1], p3[k * 3 + 2]), N = mk(n3[k * 3], n3[k * 3 + 1], n3[k * 3 + 2]);
    const LightSample ls = sample_area_light(u, r4[k * 4 + 0], r4[k * 4 + 1], P, N);
    const V3 bd = align_hemisphere(sample_cosine_hemisphere(r4[k * 4 + 2], r4[k * 4 + 3]), N);
    float *o = out + (size_t)k * 12;
    o[0] = ls.dir.x; o[1] = ls.dir.y; o[2] = ls.dir.z; o[3] = ls.dist;
    o[4] = ls.color.x; o[5] = ls.color.y; o[6] = ls.color.z; o[7] = 0.0f;
    o[8] = bd.x; o[9] = bd.y; o[10] = bd.z; o[11] = 0.0f;
}

// N1: PostProcessing.metal:44-57 -> RGBA8 (16 B read + 4 B write per pixel, HBM-streaming)
__global__ void postprocess_kernel(const v4f *accum, uint32_t w, uint32_t h, uint32_t *rgba8, int flip_y) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const uint32_t y = i / w, x = i - y * w;
    const uint32_t sy = flip_y ? (h - 1 - y) : y;
    const v4f c = accum[sy * w + x];
    const float ch[3] = { c.x, c.y, c.z };
    uint32_t packed = 0xFF000000u;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = to_srgb(aces_film(ch[a]));
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        packed |= ((uint32_t)(v * 255.0f + 0.5f)) << (8 * a);
    }
    rgba8[i] = packed;
}

__global__ void offsets_kernel(uint32_t seed, uint32_t n, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pcg_hash32(seed ^ i);
}

}  // namespace trgk_*
using namespace SFX(trgk);

namespace trg {

hipError_t SFX(launch_render)(const RenderParams &p, bool lds_scene, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    if (lds_scene) {
        if (counters) hipLaunchKernelGGL((render_kernel<true, true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
        else hipLaunchKernelGGL((render_kernel<true, false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    } else {
        if (counters) hipLaunchKernelGGL((render_kernel<false, true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
        else hipLaunchKernelGGL((render_kernel<false, false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    }
    return hipGetLastError();
}

hipError_t SFX(launch_trace)(const TraceParams &p, bool lds_scene, bool any_hit, size_t lds_bytes, hipStream_t s) {
    const dim3 grid((p.n + kBlock - 1) / kBlock);
    if (grid.x == 0) return hipSuccess;
    if (lds_scene) {
        if (any_hit) hipLaunchKernelGGL((trace_kernel<true, true>), grid, dim3(kBlock), lds_bytes, s, p);
        else hipLaunchKernelGGL((trace_kernel<true, false>), grid, dim3(kBlock), lds_bytes, s, p);
    } else {
        if (any_hit) hipLaunchKernelGGL((trace_kernel<false, true>), grid, dim3(kBlock), lds_bytes, s, p);
        else hipLaunchKernelGGL((trace_kernel<false, false>), grid, dim3(kBlock), lds_bytes, s, p);
    }
    return hipGetLastError();
}

hipError_t SFX(launch_halton)(const uint32_t *i, const uint32_t *d, uint32_t n, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(halton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, i, d, n, out);
    return hipGetLastError();
}

hipError_t SFX(launch_raygen)(const trg_uniforms &u, const uint32_t *offsets, trg_ray *out, hipStream_t s) {
    hipLaunchKernelGGL(raygen_kernel, dim3((u.width + 15) / 16, (u.height + 15) / 16), dim3(256), 0, s, u, offsets, out);
    return hipGetLastError();
}

hipError_t SFX(launch_sample)(const trg_uniforms &u, const float *p3, const float *n3, const float *r4, uint32_t n,
                              float *out12, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_kernel, dim3((n + 255) / 256), dim3(256), 0, s, u, p3, n3, r4, n, out12);
    return hipGetLastError();
}

hipError_t SFX(launch_postprocess)(const float *accum, uint32_t w, uint32_t h, uint8_t *rgba8, int flip_y, hipStream_t s) {
    const uint32_t n = w * h;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(postprocess_kernel, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const v4f *>(accum), w, h,
                       reinterpret_cast<uint32_t *>(rgba8), flip_y);
    return hipGetLastError();
}

hipError_t SFX(launch_offsets)(uint32_t seed, uint32_t n, uint32_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(offsets_kernel, dim3((n + 255) / 256), dim3(256), 0, s, seed, n, out);
    return hipGetLastError();
}

}  // namespace trg
