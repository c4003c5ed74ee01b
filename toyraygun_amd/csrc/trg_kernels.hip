// trg_kernels.hip -- the gfx950 path-tracing megakernel and its stage-level test kernels.
//
// One launch of render_kernel does what the reference does with 14 dispatches per frame x spp frames
// (src/engine/Metal/MetalRenderer.mm:400-515): raygen -> bounces x [nearest-hit -> primaryHit ->
// any-hit -> shadowHit] -> accumulate, for every frame of the requested range, with the ray, the
// throughput and the running average held in registers.  HBM traffic per pixel per launch: one
// 4-byte Halton offset read and one 16-byte float4 write (+16 B read when continuing an average).
//
// Built twice: TRG_STRICT=0 -> *_fast symbols, TRG_STRICT=1 (-ffp-contract=off) -> *_strict symbols.
#include "trg_kernels.h"
#include "trg_device.h"

using namespace trgdev;

#if TRG_STRICT
#define SFX(name) name##_strict
#else
#define SFX(name) name##_fast
#endif

#ifndef TRG_UNIFORMS_AT_USE
#define TRG_UNIFORMS_AT_USE 1
#endif
#ifndef TRG_EXPERIMENTS
#define TRG_EXPERIMENTS 0   // 1: experiments/lib/libtoyraygun_hip_exp.so -- the product library + the schedules in experiments/
#endif

// Two translation units from this one file (build.py): TRG_UNIT 1 = everything but the path-regeneration kernels, TRG_UNIT 2 = the
// path-regeneration kernels alone, in a namespace of their own (the helpers and small kernels both units contain are then different symbols)
// and compiled WITHOUT the post-RA machine scheduler (-mllvm -enable-post-misched=0): -1.8 % on C4, where the other kernels lose 0 - 0.4 % to
// that flag (profiles/r04/ab_sched_strategy.txt).  TRG_UNIT 0 (default: the developer tools) = one unit with everything.
#ifndef TRG_UNIT
#define TRG_UNIT 0
#endif
#if TRG_UNIT == 2
#define TRG_KNS SFX(trgk_regen)
#else
#define TRG_KNS SFX(trgk)
#endif

namespace TRG_KNS {

// Stage the scene blob into LDS (16-byte copies by the whole workgroup) or point at it in HBM.
template <bool LDS_SCENE>
TRG_DEV SceneView scene_view(const trg::SceneDesc &sc, unsigned char *smem) {
    SceneView v;
    if (LDS_SCENE) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.blob);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = sc.lds_stage_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += trg::kBlock) dst[i] = src[i];
        __syncthreads();
        if (kTriPlanes) {   // the shipped build tests triangles in their PLANE form: those records over the Moeller-Trumbore ones
            const uint4 *alt = reinterpret_cast<const uint4 *>(sc.blob + sc.off_tris_alt);
            uint4 *tr = reinterpret_cast<uint4 *>(smem + sc.off_tris);
            for (uint32_t i = threadIdx.x; i < sc.n_tris_rec * 3u; i += trg::kBlock) tr[i] = alt[i];
            __syncthreads();
        }
        v.nodes = reinterpret_cast<const v4f *>(smem + sc.off_nodes);
        v.tris = reinterpret_cast<const v4f *>(smem + sc.off_tris);
        v.normals = reinterpret_cast<const float *>(smem + sc.off_normals);
        v.colors = reinterpret_cast<const float *>(smem + sc.off_colors);
        v.mats = reinterpret_cast<const uint32_t *>(smem + sc.off_mats);
        v.meta = reinterpret_cast<const unsigned short *>(smem + sc.off_meta);
        v.htab = reinterpret_cast<const float *>(smem + sc.off_htab);
    } else {
        // traversed from HBM: the quantised 4-wide nodes and the 128-byte leaf records (geometry + attributes, trg_device.h kRecV4)
        v.nodes = reinterpret_cast<const v4f *>(sc.blob + (kBoxHbm ? sc.off_nodes4_box : sc.off_nodes4));   // (shipped build: the flavour whose boxes are leaves)
        v.tris = reinterpret_cast<const v4f *>(sc.blob + (kRecPlanes ? sc.off_fat_planes : sc.off_fat));
        v.normals = nullptr; v.colors = nullptr; v.mats = nullptr; v.meta = nullptr;
        v.htab = nullptr;
    }
    v.flat = reinterpret_cast<const v4f *>(sc.blob + sc.off_flat);
    v.n_flat = (LDS_SCENE && kTriPlanes) ? sc.n_flat : 0u;
    v.center = mk(sc.center[0], sc.center[1], sc.center[2]);
    v.tex.uv = nullptr; v.tex.ids = nullptr; v.tex.table = nullptr; v.tex.texels = nullptr;
    v.n_rec = sc.n_fat;
    v.rec_delta = LDS_SCENE ? 0u : (kRecPlanes ? sc.off_fat_planes : sc.off_fat) - (kBoxHbm ? sc.off_nodes4_box : sc.off_nodes4);   // (>= 0: both node arrays sit before the records, trg_capi.cpp plan_scene_layout)
    return v;
}

// lane index without touching threadIdx: v_mbcnt of an all-ones mask.  The opaque form is not hoisted or CSE'd.
TRG_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
TRG_DEV uint32_t lane_id_opaque() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
TRG_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// workgroup slot -> tile of the launch: trg_kernels.h tile_of_slot (shared with the host)
TRG_DEV bool block_tile(const trg::RenderParams &p, uint32_t slot, uint32_t &bx, uint32_t &by) {
    return trg::tile_of_slot(p.tiles_x, p.tiles_y, p.xcd_cols, slot, bx, by);
}

// storage row (row of the accumulation buffer) -> image row (trg_kernels.h RenderParams::il_n): the identity unless the launch renders
// interleaved micro-bands.  For the first row of a wavefront's sub-tile the result is wave-uniform (scalar arithmetic), and the
// sub-tile's other rows follow it (8-row micro-bands = sub-tile height).
TRG_DEV uint32_t image_row(const trg::RenderParams &p, uint32_t y_store) {
    if (p.il_n <= 1u) return y_store;
    const uint32_t l = y_store - p.row0;
    return ((l / trg::kMicroBandRows) * p.il_n + p.il_r) * trg::kMicroBandRows + (l % trg::kMicroBandRows);
}

// The shading event of one bounce (primaryHit, Raytracing.metal:115-215) for the lane's current ray and its
// nearest-hit record: updates throughput / radiance / path state, moves the ray to the continuation ray and
// returns the shadow ray to trace.  Shared by both loop shapes of render_kernel.
struct ShadeOut { bool want_shadow, want_next, shaded; V3 sdir, scol; float smax; };
template <bool TAB = false, bool FAT = false>
TRG_DEV ShadeOut shade_event(const trg_uniforms &u, const SceneView &sc, const Hit &h, bool found, uint32_t b, bool last, uint32_t hidx,
                             V3 &o, V3 &d, V3 &thr, V3 &rad, uint32_t &rmask, bool &active, V3 light_color, const float *rpre = nullptr) {
    ShadeOut out;
    out.want_shadow = false; out.want_next = false; out.shaded = false; out.sdir = mk(0.0f, 0.0f, 1.0f); out.scol = mk(0.0f, 0.0f, 0.0f); out.smax = -1.0f;
    if (!found) {
        active = false;  // Raytracing.metal:139-144
        return out;
    }
    const Surf<FAT> sf = surf_fetch<FAT>(sc, h.prim);   // FAT: h.prim is the leaf record of the hit
    const uint32_t mat = sf.mat;
    if (mat == TRG_MATERIAL_DEFAULT) {
        out.shaded = true;
        // Raytracing.metal:150-199
        const V3 P = o + d * h.t;
        const float cx = 1.0f - h.u - h.v, cy = h.u;  // weights of vertex 0, 1
        V3 vcol, nraw;
        surf_interp<FAT>(sc, sf, cx, cy, vcol, nraw);
        const V3 nrm = normalize(nraw);
        float r[4];
        // opaque copy: stops LICM from hoisting every bounce's Halton digits (all 60 dimensions) out of the
        // bounce loop and keeping them live in VGPRs
        uint32_t hi = hidx;
        asm volatile("" : "+v"(hi));
        r[2] = 0.0f; r[3] = 0.0f;
        if (rpre) { r[0] = rpre[0]; r[1] = rpre[1]; r[2] = rpre[2]; r[3] = rpre[3]; }   // the caller evaluated this bounce's dimensions
        else if (last) halton2<TAB>(hi, b, r, sc.htab);
        else halton4<TAB>(hi, b, r, sc.htab);
        const LightSample ls = sample_area_light(u, r[0], r[1], P, nrm);
        thr = thr * vcol;
        o = P + nrm * 1e-3f;  // origin of both the shadow ray and the continuation ray
        out.smax = ls.dist - 1e-3f;
        out.sdir = ls.dir;
        out.scol = ls.color * thr;
        out.want_shadow = out.smax >= 0.0f;  // inactive shadow rays are not traced (MPS skips maxDistance < 0)
        if (!last) {  // the reference also writes a continuation ray on the last bounce; nothing reads it
            d = align_hemisphere(sample_cosine_hemisphere(r[2], r[3]), nrm);
            out.want_next = true;
        }
        rmask = 1u;  // RAY_MASK_SECONDARY
    } else if (mat == TRG_MATERIAL_EMISSIVE) {
        rad = light_color;  // Raytracing.metal:200-209: overwrites the texel, ends the path
        active = false;
    } else {
        rad = mk(1.0f, 0.0f, 1.0f);  // Raytracing.metal:210-214: magenta, ray left as is (traced again)
        out.want_next = !last;
    }
    return out;
}

// One path: raygen, then `bounces` x [nearest hit, shading event, shadow ray] -- the radiance this frame adds to the
// pixel's texel (what the reference's render target holds before accumulate).  Shared by the frame-serial and the
// frame-parallel megakernels.
// Ray counters are WAVE-level scalars: every count is taken at a wave-uniform point as popcount(ballot(predicate)), so the four
// counters live in SGPRs instead of four VGPRs per lane (the per-lane form cost 4 of the 64 registers of the 8-waves/SIMD build).
struct PathCounters { uint32_t primary, bounce, shadow, shaded; };
TRG_DEV uint32_t wave_count(bool pred) { return (uint32_t)__popcll(__ballot(pred)); }
typedef __attribute__((address_space(3))) float lds_float_t;
template <bool LDS_SCENE, bool COUNT, typename STK>
TRG_DEV V3 path_radiance(const trg::RenderParams &p, const SceneView &sc, STK stk, uint32_t x, uint32_t y, uint32_t offset, uint32_t frame, bool valid,
                         V3 light_color, PathCounters &pc, Counters &cnt, lds_float_t *path_park = nullptr) {
    // Halton index of this pixel-sample (Raytracing.metal:67: offset + uniforms.frameIndex, wraps mod 2^32).  `frame` is wave-uniform:
    // the sum is re-formed where it is used instead of living in a VGPR of its own across the traversals.
    // (HBM scenes: the offset itself is parked in LDS behind the path state and read back where the index is formed)
#define TRG_HIDX ((TRG_PARK_OFFSET && path_park ? (uint32_t)__float_as_int(path_park[6 * trg::kBlock]) : offset) + frame)
    V3 o, d;
    constexpr bool TAB = LDS_SCENE && !TRG_STRICT && TRG_HALTON_TABLES;  // Halton group tables staged with the scene
    // The 176-byte uniform block is read from the kernel-argument segment WHERE IT IS USED (scalar loads, cached) through a pointer
    // the optimiser cannot see through: loaded once at kernel entry its 44 dwords sat in SGPRs across every traversal, and with
    // 80 (LDS scenes) / 96 (HBM scenes) SGPRs per wave about 60 of them were spilled to VGPR lanes and, from there, to scratch.
#if TRG_UNIFORMS_AT_USE
    typedef const __attribute__((address_space(4))) trg_uniforms cu_t;
    cu_t *up = (cu_t *)__builtin_amdgcn_kernarg_segment_ptr();   // RenderParams::u is the first member
    static_assert(offsetof(trg::RenderParams, u) == 0, "uniforms must lead the kernel arguments");
    asm volatile("" : "+s"(up));
#define TRG_U (*(const trg_uniforms *)up)
#else
#define TRG_U p.u
#endif
    raygen<TAB>(TRG_U, x, y, TRG_HIDX, o, d, sc.htab);
    V3 thr = mk(1.0f, 1.0f, 1.0f);  // ray.color
    V3 rad = mk(0.0f, 0.0f, 0.0f);  // the render target texel of this frame
    // the ray mask is RAY_MASK_PRIMARY (3) until the first shading event and RAY_MASK_SECONDARY (1) afterwards (a path that only ever
    // hit an invalid material keeps its primary ray): one bit per lane -- a lane mask in SGPRs -- instead of a VGPR
    bool primary_ray = true;
#define TRG_RMASK (primary_ray ? 3u : 1u)
    bool active = valid;
    pc.primary += wave_count(active);

    if (LDS_SCENE) {
        // LDS-resident scene: per bounce, a wave-synchronous nearest-hit trace, the shading event, a
        // wave-synchronous any-hit trace.
        for (uint32_t b = 0; b < p.bounces; ++b) {
            if (__ballot(active) == 0ull) break;  // whole wavefront terminated
            const bool last = (b + 1u == p.bounces);  // wave-uniform
#if TRG_UNIFORMS_AT_USE
            asm volatile("" : "+s"(up));   // the uniforms of this bounce's shading event are loaded after this point
#endif
            if (b > 0) pc.bounce += wave_count(active);
            ShadeOut so; so.want_shadow = false; so.shaded = false;
            if (active) {
                Hit h;
                const bool found = traverse<false, COUNT, trg::kBlock, false>(sc, o, d, INFINITY, TRG_RMASK, h, stk, cnt);
                uint32_t rmask = TRG_RMASK;
                so = shade_event<TAB>(TRG_U, sc, h, found, b, last, TRG_HIDX, o, d, thr, rad, rmask, active, light_color);
                primary_ray = rmask == 3u;
                if (so.want_shadow) {
                    Hit sh;
                    const bool occluded = traverse<true, COUNT, trg::kBlock, false>(sc, o, so.sdir, so.smax, 1u, sh, stk, cnt);
                    if (!occluded) rad = rad + so.scol;  // Raytracing.metal:240-241
                }
            }
            pc.shaded += wave_count(so.shaded);
            pc.shadow += wave_count(so.want_shadow);
        }
    } else {
        // HBM-resident scene: the primary ray alone; afterwards every shading event yields a (shadow ray,
        // continuation ray) pair that the lane traces back to back in one loop (traverse_pair): +5 % on C4.
        // (On the LDS-resident Cornell box pairing costs 7 %: the bounce-0 shadow rays are coherent and cheap,
        // and the incoherent continuation rays give the wavefront nothing to overlap them with.)
        Hit h; h.t = -1.0f; h.prim = -1; h.u = 0.0f; h.v = 0.0f;
        bool found = false;
        if (p.bounces > 0u && active) found = traverse<false, COUNT, trg::kBlock, true>(sc, o, d, INFINITY, TRG_RMASK, h, stk, cnt);
        for (uint32_t b = 0; b < p.bounces; ++b) {
            if (__ballot(active) == 0ull) break;
            const bool last = (b + 1u == p.bounces);
#if TRG_UNIFORMS_AT_USE
            asm volatile("" : "+s"(up));
#endif
            ShadeOut so; so.want_shadow = false; so.want_next = false; so.shaded = false; so.sdir = mk(0.0f, 0.0f, 1.0f); so.scol = mk(0.0f, 0.0f, 0.0f); so.smax = -1.0f;
            if (active) {
                uint32_t rmask = TRG_RMASK;
                so = shade_event<TAB, true>(TRG_U, sc, h, found, b, last, TRG_HIDX, o, d, thr, rad, rmask, active, light_color);
                primary_ray = rmask == 3u;
            }
            pc.shaded += wave_count(so.shaded);
            pc.shadow += wave_count(so.want_shadow);
            pc.bounce += wave_count(so.want_next);
            if (__ballot(so.want_shadow || so.want_next) != 0ull) {
                bool occluded = false;
                // throughput and radiance are not needed while the pair is traced: parked in LDS ([component][thread]) instead of
                // being spilled to scratch by the register allocator (6 scratch stores + 6 loads per bounce in profiles/r01)
                if (path_park) {
                    path_park[0] = thr.x; path_park[trg::kBlock] = thr.y; path_park[2 * trg::kBlock] = thr.z;
                    path_park[3 * trg::kBlock] = rad.x; path_park[4 * trg::kBlock] = rad.y; path_park[5 * trg::kBlock] = rad.z;
                }
                traverse_pair<COUNT, trg::kBlock, true>(sc, o, so.want_shadow, so.sdir, so.smax, so.want_next, d, TRG_RMASK, occluded, h, found, stk, cnt);
                if (path_park) {
                    thr = mk(path_park[0], path_park[trg::kBlock], path_park[2 * trg::kBlock]);
                    rad = mk(path_park[3 * trg::kBlock], path_park[4 * trg::kBlock], path_park[5 * trg::kBlock]);
                }
                if (so.want_shadow && !occluded) rad = rad + so.scol;
            }
        }
    }
    return rad;
#undef TRG_HIDX
#undef TRG_RMASK
#undef TRG_U
}

// minimum waves per SIMD the register allocator must leave room for.  Scene in LDS: 8 (at most 64 VGPRs; VALU-bound,
// more waves hide the LDS latency: +3 % C2, +5 % at 8 bounces over 6).  Scene in HBM: 7 (at most 72 VGPRs; 8 spills
// and is 40 % slower, 6 loses 5 % of latency hiding; with the quantised 64-byte nodes 7 = 72 VGPRs is the sweet spot).
#ifndef TRG_EXP_WAVES
#define TRG_EXP_WAVES 8
#endif
#ifndef TRG_EXP_WAVES_HBM
#define TRG_EXP_WAVES_HBM 7
#endif
template <bool LDS_SCENE, bool COUNT>
__global__ __launch_bounds__(trg::kBlock, LDS_SCENE ? TRG_EXP_WAVES : TRG_EXP_WAVES_HBM) void render_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SceneView sc = scene_view<LDS_SCENE>(p.sc, smem);
    sc.tex = p.tex;
    LdsStackT<trg::kBlock, !LDS_SCENE> stk;  // HBM scenes may spill deep stack levels to global scratch
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);  // also writes the sentinel at level 0 of this thread's column

    // workgroup -> 16x16 tile, wavefront -> 8x8 sub-tile, lane -> pixel (8 consecutive pixels of a row
    // per 8 lanes: each wavefront writes eight 128-byte row segments)
    // The wavefront index is wave-uniform (an SGPR) and the lane index is re-read with v_mbcnt wherever it is needed, so that
    // pixel coordinates, the pixel index and threadIdx do not occupy VGPRs across the frame loop (they used to be spilled).
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t bx, by;
    if (!block_tile(p, blockIdx.x, bx, by)) return;   // (a padding slot of the XCD-aware order)
    const uint32_t x0 = bx * trg::kTileW + (wave % (trg::kTileW / 8)) * 8;           // wave-uniform
    const uint32_t y0 = p.row0 + by * trg::kTileH + (wave / (trg::kTileW / 8)) * 8;  // wave-uniform: row of the accumulation buffer
    const uint32_t y0i = image_row(p, y0);                                            // wave-uniform: row of the image (interleaved bands)
    bool valid;
    uint32_t offset = 0u;
    v4f *accum = reinterpret_cast<v4f *>(p.accum);
    // The running average of the pixel: in registers for an LDS-resident scene; PARKED IN LDS between frames for an HBM-resident
    // one (its traversal needs the registers, and its LDS holds only the stacks): [component][thread], conflict-free.
    constexpr bool PARK = !LDS_SCENE;
    lds_float_t *park = (lds_float_t *)(reinterpret_cast<float *>(smem + p.acc_off) + threadIdx.x);
    lds_float_t *path_park = (PARK && TRG_PARK_PATH) ? park + 3 * trg::kBlock : nullptr;
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    {
        const uint32_t lane = lane_id();
        const uint32_t x = x0 + (lane & 7), y = y0 + (lane >> 3), yi = y0i + (lane >> 3);
        valid = (x < p.u.width) && (y < p.row0 + p.rows) && (yi < p.u.height);
        const uint32_t pix = y * p.u.width + x;
        if (valid) offset = p.offsets[yi * p.u.width + x];
        if (valid && p.frame_begin > 0) { const v4f a = accum[pix]; acc = mk(a.x, a.y, a.z); }
        if (PARK) { park[0] = acc.x; park[trg::kBlock] = acc.y; park[2 * trg::kBlock] = acc.z; }
        if (TRG_PARK_OFFSET && path_park) { path_park[6 * trg::kBlock] = __int_as_float((int)offset); offset = 0u; }
    }

    PathCounters pc; pc.primary = 0; pc.bounce = 0; pc.shadow = 0; pc.shaded = 0;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);

    for (uint32_t f = p.frame_begin; f < p.frame_begin + p.spp; ++f) {
        // the lane index is re-read every frame (an opaque v_mbcnt pair): otherwise LICM hoists the pixel coordinates and
        // raygen's per-pixel products out of the frame loop, values that then live in VGPRs across every traversal of every
        // frame and were spilled to scratch at 8 waves/SIMD (profiles/r01: 13 scratch stores per pixel = the 3.4x write
        // amplification, 8 scratch reloads per frame).  Recomputing them costs a dozen VALU instructions per frame.
        const uint32_t lane_f = lane_id_opaque();
        const uint32_t xf = x0 + (lane_f & 7), yf = y0i + (lane_f >> 3);
        const V3 rad = path_radiance<LDS_SCENE, COUNT>(p, sc, stk, xf, yf, offset, f, valid, light_color, pc, cnt, path_park);
        if (PARK) acc = mk(park[0], park[trg::kBlock], park[2 * trg::kBlock]);
        // Accumulate.metal:19-39
        if (f == 0) {
            acc = rad;
        } else {
            const V3 prev = acc * (float)f;
            const V3 c = rad + prev;
            const float f1 = (float)(f + 1u);
            acc = mk(c.x / f1, c.y / f1, c.z / f1);
        }
        if (PARK) { park[0] = acc.x; park[trg::kBlock] = acc.y; park[2 * trg::kBlock] = acc.z; }
    }
    const uint32_t lane = lane_id_opaque();
    if (valid) {
        const uint32_t x = x0 + (lane & 7), y = y0 + (lane >> 3);
        if (PARK) acc = mk(park[0], park[trg::kBlock], park[2 * trg::kBlock]);
        v4f outv; outv.x = acc.x; outv.y = acc.y; outv.z = acc.z; outv.w = 1.0f;
        accum[y * p.u.width + x] = outv;
    }

    // ray counters: wavefront reduce -> workgroup reduce in LDS -> one atomic per counter per workgroup,
    // spread over kCounterSlots slots
    uint32_t vals[8] = { pc.primary, pc.bounce, pc.shadow, pc.shaded, cnt.nodes, cnt.tris, cnt.wnodes, cnt.wtris };
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + p.red_off);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!COUNT && k >= 4) break;
        const uint32_t s = k < 4 ? vals[k] : wave_sum(vals[k]);   // the ray counters are already wave totals
        if (lane == 0) red[wave * 8 + k] = s;
    }
    __syncthreads();
    if (wave == 0 && lane < (COUNT ? 8u : 4u)) {
        const uint32_t k = lane;
        unsigned long long s = 0;
        for (int wv = 0; wv < trg::kWaves; ++wv) s += red[wv * 8 + k];
        if (s) atomicAdd(&p.counters[(blockIdx.x % trg::kCounterSlots) * trg::kCounterWords + k], s);
    }
}

#include "trg_regen.inc.h"

// ---------------------------------------------------------------------------------------------
// render_fp_kernel: the frame-parallel schedule of the same megakernel, for launches whose pixel grid alone does
// not fill the chip (a row band of a multi-GPU job, a small window).  render_kernel gives a pixel to ONE lane that
// walks its `spp` frames one after the other, so a launch of fewer workgroups than the 256 CUs can hold takes as
// long as that serial chain however idle the machine is.  Here a workgroup covers 4/F 8x8 sub-tiles side by side
// and F "frame lanes": wave (sub, fl) traces frames fl, fl+F, fl+2F ... of its sub-tile and parks each frame's
// radiance in LDS; after every R rounds the fl=0 wave of each sub-tile folds the parked radiances into the running
// average IN FRAME ORDER -- the same float operations in the same order as Accumulate.metal, so the result is
// bit-identical to render_kernel's (strict build tested bit-exact against the oracle and against render_kernel).
// F x as many workgroups, each with 1/F of the serial work.
// ---------------------------------------------------------------------------------------------
#ifndef TRG_EXP_WAVES_FP
#define TRG_EXP_WAVES_FP 6   // its grids do not fill the chip anyway: 80 VGPRs without spills beats 8 waves by 1.5 %
#endif
#ifndef TRG_EXP_WAVES_FP_HBM
#define TRG_EXP_WAVES_FP_HBM 6   // HBM scene: 79 VGPRs and no scratch with throughput / radiance / offset parked in LDS (round 2: 72 VGPRs, 31 spilled, 96 B/lane at 7)
#endif
template <bool LDS_SCENE, bool COUNT>
__global__ __launch_bounds__(trg::kBlock, LDS_SCENE ? TRG_EXP_WAVES_FP : TRG_EXP_WAVES_FP_HBM) void render_fp_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SceneView sc = scene_view<LDS_SCENE>(p.sc, smem);
    sc.tex = p.tex;
    LdsStackT<trg::kBlock, !LDS_SCENE> stk;
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);  // also writes the sentinel at level 0 of this thread's column

    // wave-uniform indices in SGPRs, the lane index re-read where it is needed (as render_kernel: pixel coordinates and indices must not
    // live in VGPRs across the traversals)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t F = p.fsplit, subt = trg::kWaves / F;   // frame lanes, sub-tiles per workgroup (F in {2,4})
    const uint32_t sub = wave % subt, fl = wave / subt;
    uint32_t bx, by;                                         // tile order as render_kernel
    if (!block_tile(p, blockIdx.x, bx, by)) return;
    const uint32_t x0 = (bx * subt + sub) * 8u, y0 = p.row0 + by * 8u;   // wave-uniform
    const uint32_t y0i = image_row(p, y0);                                // row of the image (interleaved bands)
    bool valid;
    uint32_t offset = 0u;
    v4f *accum = reinterpret_cast<v4f *>(p.accum);
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    // an HBM-resident scene parks throughput, radiance and the pixel's Halton offset in LDS while rays are traced (path_radiance), as
    // render_kernel does: seven words per thread behind the parked radiances
    constexpr bool PARK = !LDS_SCENE && TRG_PARK_PATH && TRG_PARK_OFFSET;
    lds_float_t *path_park = PARK ? (lds_float_t *)(reinterpret_cast<float *>(smem + p.acc_off) + threadIdx.x) : nullptr;
    {
        const uint32_t lane = lane_id();
        const uint32_t x = x0 + (lane & 7), y = y0 + (lane >> 3), yi = y0i + (lane >> 3);
        valid = (x < p.u.width) && (y < p.row0 + p.rows) && (yi < p.u.height);
        const uint32_t pix = y * p.u.width + x;
        if (valid) offset = p.offsets[yi * p.u.width + x];
        if (fl == 0 && valid && p.frame_begin > 0) { const v4f a = accum[pix]; acc = mk(a.x, a.y, a.z); }
        if (PARK) { path_park[6 * trg::kBlock] = __int_as_float((int)offset); offset = 0u; }
    }

    PathCounters pc; pc.primary = 0; pc.bounce = 0; pc.shadow = 0; pc.shaded = 0;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);

    // parked radiance: slot-major, component, then the 64*subt pixels of the workgroup (conflict-free rows)
    float *park = reinterpret_cast<float *>(smem + p.pool_off);
    const uint32_t npx = 64u * subt;
    const uint32_t chunk = p.fp_rounds * F;
    for (uint64_t rel = 0; rel < p.spp; rel += chunk) {
        for (uint32_t r = 0; r < p.fp_rounds; ++r) {
            const uint64_t i = rel + (uint64_t)r * F + fl;
            if (i >= p.spp) break;   // wave-uniform
            const uint32_t lane_f = lane_id_opaque();
            const V3 rad = path_radiance<LDS_SCENE, COUNT>(p, sc, stk, x0 + (lane_f & 7), y0i + (lane_f >> 3), offset, p.frame_begin + (uint32_t)i, valid, light_color, pc, cnt, path_park);
            float *slot = park + (size_t)(r * F + fl) * 3u * npx + sub * 64u + lane_id_opaque();
            slot[0] = rad.x; slot[npx] = rad.y; slot[2u * npx] = rad.z;
        }
        __syncthreads();
        if (fl == 0) {
            const uint32_t mypx = sub * 64u + lane_id_opaque();
            for (uint32_t s = 0; s < chunk && rel + s < p.spp; ++s) {
                const uint32_t f = p.frame_begin + (uint32_t)(rel + s);
                const float *slot = park + (size_t)s * 3u * npx + mypx;
                const V3 rad = mk(slot[0], slot[npx], slot[2u * npx]);
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
        }
        __syncthreads();
    }
    const uint32_t lane = lane_id_opaque();
    if (fl == 0 && valid) {
        const uint32_t x = x0 + (lane & 7), y = y0 + (lane >> 3);
        v4f outv; outv.x = acc.x; outv.y = acc.y; outv.z = acc.z; outv.w = 1.0f;
        accum[y * p.u.width + x] = outv;
    }

    uint32_t vals[8] = { pc.primary, pc.bounce, pc.shadow, pc.shaded, cnt.nodes, cnt.tris, cnt.wnodes, cnt.wtris };
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + p.red_off);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!COUNT && k >= 4) break;
        const uint32_t s = k < 4 ? vals[k] : wave_sum(vals[k]);   // the ray counters are already wave totals
        if (lane == 0) red[wave * 8 + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < (COUNT ? 8 : 4)) {
        const uint32_t k = threadIdx.x;
        unsigned long long s = 0;
        for (int wv = 0; wv < trg::kWaves; ++wv) s += red[wv * 8 + k];
        if (s) atomicAdd(&p.counters[(blockIdx.x % trg::kCounterSlots) * trg::kCounterWords + k], s);
    }
}

// The schedules that were built, measured and lost -- the workgroup path pool (render_pool_kernel) and the wavefront schedule (wf_* kernels) --
// live in experiments/ and are compiled into experiments/lib/libtoyraygun_hip_exp.so only (-DTRG_EXPERIMENTS=1; experiments/README.md).
#if TRG_EXPERIMENTS
#include "../../experiments/trg_pool.inc.h"
#include "../../experiments/trg_wavefront.inc.h"
#endif
#include "trg_tail.inc.h"

// ---- a7 / a12 in isolation: one ray per thread ----
template <bool LDS_SCENE, bool ANY>
__global__ __launch_bounds__(trg::kBlock) void trace_kernel(const trg::TraceParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SceneView sc = scene_view<LDS_SCENE>(p.sc, smem);
    LdsStackT<trg::kBlock, !LDS_SCENE> stk;  // HBM scenes may spill deep stack levels to global scratch
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);  // also writes the sentinel at level 0 of this thread's column
    const uint32_t i = blockIdx.x * trg::kBlock + threadIdx.x;
    if (i >= p.n) return;
    const trg_ray r = p.rays[i];
    const V3 o = mk(r.origin[0], r.origin[1], r.origin[2]), d = mk(r.direction[0], r.direction[1], r.direction[2]);
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
    Hit h; h.t = -1.0f; h.prim = -1; h.u = 0.0f; h.v = 0.0f;
    bool found = false;
    if (r.maxDistance >= 0.0f) found = traverse<ANY, false, trg::kBlock, !LDS_SCENE>(sc, o, d, r.maxDistance, r.mask, h, stk, cnt);
    if (ANY) {
        reinterpret_cast<float *>(p.out)[i] = found ? h.t : -1.0f;
    } else {
        trg_isect is;
        is.distance = found ? h.t : -1.0f;
        // an HBM-resident scene names the leaf record of the hit: its first row carries the original index
        is.primitiveIndex = found ? (LDS_SCENE ? h.prim : fat_prim(sc.tris, (uint32_t)h.prim)) : -1;
        is.coordinates[0] = found ? (1.0f - h.u - h.v) : 0.0f;
        is.coordinates[1] = found ? h.u : 0.0f;
        reinterpret_cast<trg_isect *>(p.out)[i] = is;
    }
}

__global__ void halton_kernel(const uint32_t *i, const uint32_t *d, uint32_t n, float *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = halton_any(i[k], d[k]);
}

// a5 as the SHIPPED megakernel evaluates it on an LDS-resident scene: the group tables staged into LDS exactly like
// scene_view does, dimensions 1..5 through halton_t (checked), the others through the digit code
__global__ void halton_tab_kernel(const float *htab_global, const uint32_t *i, const uint32_t *d, uint32_t n, float *out) {
    __shared__ float T[trg::kHtabBytes / 4];
    for (uint32_t k = threadIdx.x; k < trg::kHtabBytes / 4; k += blockDim.x) T[k] = htab_global[k];
    __syncthreads();
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = halton_any_tab(i[k], d[k], T);
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

// interleaved row bands -> image rows (trg_unpack_bands): the gathered COMPACT frame holds rank r's micro-bands r, r + n, r + 2n ... in
// rows [r * B, r * B + its rows); image row y = micro-band y / 8 belongs to rank (y / 8) % n, local micro-band (y / 8) / n.
// 16 B read + 16 B write per pixel, HBM-streaming.
__global__ void unpack_bands_kernel(const v4f *compact, v4f *image, uint32_t w, uint32_t h, uint32_t n, uint32_t B) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const uint32_t y = i / w, x = i - y * w;
    const uint32_t mb = y / trg::kMicroBandRows, r = mb % n, l = (mb / n) * trg::kMicroBandRows + y % trg::kMicroBandRows;
    image[i] = compact[(size_t)(r * B + l) * w + x];
}

// which XCD does workgroup b run on?  (HW_REG_XCC_ID, bits 3:0; MI355X_MICROARCH.md "Workgroup dispatch, XCD placement")
__global__ void xcc_probe_kernel(uint32_t *out) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = id & 15u;
}

__global__ void offsets_kernel(uint32_t seed, uint32_t n, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pcg_hash32(seed ^ i);
}

}  // namespace trgk_*
using namespace TRG_KNS;

namespace trg {

#if TRG_UNIT != 2
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

#endif  // TRG_UNIT != 2
#if TRG_UNIT != 1
hipError_t SFX(launch_render_regen)(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    if (p.xq) {   // persistent workgroups popping from per-XCD job queues (TRG_OPT_TILE_ORDER 64 + n)
        if (counters) hipLaunchKernelGGL((render_regen_kernel<true, true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
        else hipLaunchKernelGGL((render_regen_kernel<false, true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
        return hipGetLastError();
    }
    if (counters) hipLaunchKernelGGL((render_regen_kernel<true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    else hipLaunchKernelGGL((render_regen_kernel<false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    return hipGetLastError();
}

hipError_t SFX(launch_regen_accumulate)(const RenderParams &p, uint32_t grid, hipStream_t s) {
    hipLaunchKernelGGL(regen_accumulate_kernel, dim3(grid), dim3(kBlock), (size_t)p.spp * kBlock * 16u, s, p);
    return hipGetLastError();
}
#endif  // TRG_UNIT != 1
#if TRG_UNIT != 2

template <typename K>
static hipError_t launch_big_lds(K kernel, const RenderParams &p, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    // more than 64 KB of dynamic LDS has to be opted into per kernel (gfx950 has 160 KB per CU)
    if (lds_bytes > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds_bytes, s, p);
    return hipGetLastError();
}

#if TRG_EXPERIMENTS
hipError_t SFX(launch_render_pool)(const RenderParams &p, bool lds_scene, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    constexpr int S = kPoolS;
    if (lds_scene) {
        if (counters) return launch_big_lds(render_pool_kernel<true, true, S>, p, grid, lds_bytes, s);
        return launch_big_lds(render_pool_kernel<true, false, S>, p, grid, lds_bytes, s);
    }
    if (counters) return launch_big_lds(render_pool_kernel<false, true, S>, p, grid, lds_bytes, s);
    return launch_big_lds(render_pool_kernel<false, false, S>, p, grid, lds_bytes, s);
}

#endif  // TRG_EXPERIMENTS

hipError_t SFX(launch_render_fp)(const RenderParams &p, bool lds_scene, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    if (lds_scene) {
        if (counters) return launch_big_lds(render_fp_kernel<true, true>, p, grid, lds_bytes, s);
        return launch_big_lds(render_fp_kernel<true, false>, p, grid, lds_bytes, s);
    }
    if (counters) return launch_big_lds(render_fp_kernel<false, true>, p, grid, lds_bytes, s);
    return launch_big_lds(render_fp_kernel<false, false>, p, grid, lds_bytes, s);
}

hipError_t SFX(launch_render_head)(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    if (counters) hipLaunchKernelGGL((render_head_kernel<true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    else hipLaunchKernelGGL((render_head_kernel<false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    return hipGetLastError();
}
hipError_t SFX(launch_render_tail)(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    if (counters) hipLaunchKernelGGL((render_tail_kernel<true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    else hipLaunchKernelGGL((render_tail_kernel<false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    return hipGetLastError();
}
#if TRG_EXPERIMENTS
hipError_t SFX(launch_render_rtail)(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    if (counters) hipLaunchKernelGGL((render_rtail_kernel<true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    else hipLaunchKernelGGL((render_rtail_kernel<false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    return hipGetLastError();
}
#endif
hipError_t SFX(launch_tail_sort)(const RenderParams &p, uint32_t grid, uint32_t mode, const float *lo3, const float *inv3, hipStream_t s) {
    hipLaunchKernelGGL(tail_sort_kernel, dim3(grid), dim3(256), 0, s, p, mode, lo3[0], lo3[1], lo3[2], inv3[0], inv3[1], inv3[2]);
    return hipGetLastError();
}
hipError_t SFX(launch_tail_accumulate)(const RenderParams &p, hipStream_t s) {
    hipLaunchKernelGGL(tail_accumulate_kernel, dim3((p.tail_band_pixels + 255u) / 256u), dim3(256), 0, s, p);
    return hipGetLastError();
}
#if TRG_EXPERIMENTS
hipError_t SFX(launch_wf_raygen)(const WfParams &p, hipStream_t s) {
    hipLaunchKernelGGL(wf_raygen_kernel, dim3((p.nb + 255u) / 256u), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t SFX(launch_wf_trace)(const WfParams &p, bool lds_scene, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s) {
    if (lds_scene) {
        if (counters) hipLaunchKernelGGL((wf_trace_kernel<true, true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
        else hipLaunchKernelGGL((wf_trace_kernel<true, false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    } else {
        if (counters) hipLaunchKernelGGL((wf_trace_kernel<false, true>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
        else hipLaunchKernelGGL((wf_trace_kernel<false, false>), dim3(grid), dim3(kBlock), lds_bytes, s, p);
    }
    return hipGetLastError();
}
hipError_t SFX(launch_wf_shade)(const WfParams &p, bool lds_scene, uint32_t grid, hipStream_t s) {
    if (lds_scene) hipLaunchKernelGGL((wf_shade_kernel<false>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wf_shade_kernel<true>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t SFX(launch_wf_accumulate)(const WfParams &p, hipStream_t s) {
    hipLaunchKernelGGL(wf_accumulate_kernel, dim3((p.npix + 255u) / 256u), dim3(256), 0, s, p);
    return hipGetLastError();
}
#endif  // TRG_EXPERIMENTS

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

hipError_t SFX(launch_halton_tab)(const float *htab, const uint32_t *i, const uint32_t *d, uint32_t n, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(halton_tab_kernel, dim3((n + 255) / 256), dim3(256), 0, s, htab, i, d, n, out);
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

hipError_t SFX(launch_unpack_bands)(const float *compact, float *image, uint32_t w, uint32_t h, uint32_t n, uint32_t B, hipStream_t s) {
    const uint32_t px = w * h;
    if (px == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_bands_kernel, dim3((px + 255) / 256), dim3(256), 0, s, reinterpret_cast<const v4f *>(compact), reinterpret_cast<v4f *>(image), w, h, n, B);
    return hipGetLastError();
}

hipError_t SFX(launch_xcc_probe)(uint32_t n_blocks, uint32_t *out, hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(n_blocks), dim3(256), 0, s, out);
    return hipGetLastError();
}

hipError_t SFX(launch_offsets)(uint32_t seed, uint32_t n, uint32_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(offsets_kernel, dim3((n + 255) / 256), dim3(256), 0, s, seed, n, out);
    return hipGetLastError();
}

#endif  // TRG_UNIT != 2

}  // namespace trg
