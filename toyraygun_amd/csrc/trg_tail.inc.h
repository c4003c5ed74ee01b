// trg_tail.inc.h -- TAIL COMPACTION for deep paths on an LDS-resident scene (TRG_OPT_TAIL_BOUNCE), included by trg_kernels.hip inside
// namespace trgk_{fast,strict}.
//
// Why.  The direct megakernel runs every bounce of a frame for all 64 lanes of a wavefront, whatever is left of their paths: a
// bounce costs the wavefront the same time with 60 live lanes or with 9.  On the closed Cornell box 72 % of the paths survive each
// bounce after the first, so at 8 bounces (C3) the stages run at 100, 98, 72, 52, 37, 27, 19, 14 % live lanes -- profiles/r02: 15.5 of
// 64 lanes active in the node loop.  Here the megakernel stops after bounce K-1 ("head"); a wavefront appends the paths that go on
// -- origin, direction, throughput, pixel, frame: 48 bytes -- to ITS OWN segment of a queue in HBM with a ballot / mbcnt prefix (no
// atomics: the fill level is a wave-uniform register, written once at the end), and a second launch ("tail") runs bounces K.. for
// 64 consecutive queue entries per wavefront: every lane starts with a live path again.  The frame's radiance travels through a
// per-frame buffer (the tail continues the same sequence of float additions the head began), and a third kernel folds the frames
// of the chunk into the running average in frame order (Accumulate.metal:19-39): bit-identical to render_kernel.
//
// Layout: frames are processed in chunks of Fc <= 16.  queue[W * cap + k], cap = 64 * Fc, W = head wavefront (4 per workgroup),
// entry = 3 float4: (o.xyz, bits pixel-in-band) (d.xyz, bits frame-in-chunk) (thr.xyz, 0); count[W]; radbuf[frame-in-chunk][pixel-in-band].

// bounces [b0, b1) of one path per lane on an LDS-resident scene: the loop of path_radiance, re-entrant
template <bool COUNT, typename STK>
TRG_DEV void path_segment_lds(const trg::RenderParams &p, const SceneView &sc, STK stk, uint32_t offset, uint32_t frame, uint32_t b0, uint32_t b1,
                              V3 &o, V3 &d, V3 &thr, V3 &rad, bool &active, bool &primary_ray, V3 light_color, PathCounters &pc, Counters &cnt) {
    constexpr bool TAB = !TRG_STRICT && TRG_HALTON_TABLES;
    typedef const __attribute__((address_space(4))) trg_uniforms cu_t;
    cu_t *up = (cu_t *)__builtin_amdgcn_kernarg_segment_ptr();   // RenderParams::u is the first member (see path_radiance)
    for (uint32_t b = b0; b < b1; ++b) {
        if (__ballot(active) == 0ull) break;
        const bool last = (b + 1u == p.bounces);
        asm volatile("" : "+s"(up));
        if (b > 0) pc.bounce += wave_count(active);
        ShadeOut so; so.want_shadow = false; so.shaded = false;
        if (active) {
            Hit h;
            const bool found = traverse<false, COUNT, trg::kBlock, false>(sc, o, d, INFINITY, primary_ray ? 3u : 1u, h, stk, cnt);
            uint32_t rmask = primary_ray ? 3u : 1u;
            so = shade_event<TAB>(*(const trg_uniforms *)up, sc, h, found, b, last, offset + frame, o, d, thr, rad, rmask, active, light_color);
            primary_ray = rmask == 3u;
            if (so.want_shadow) {
                Hit sh;
                const bool occluded = traverse<true, COUNT, trg::kBlock, false>(sc, o, so.sdir, so.smax, 1u, sh, stk, cnt);
                if (!occluded) rad = rad + so.scol;
            }
        }
        pc.shaded += wave_count(so.shaded);
        pc.shadow += wave_count(so.want_shadow);
    }
}

TRG_DEV void flush_counters(const trg::RenderParams &p, unsigned char *smem, uint32_t wave, uint32_t lane, const PathCounters &pc, const Counters &cnt, bool count) {
    uint32_t vals[8] = { pc.primary, pc.bounce, pc.shadow, pc.shaded, cnt.nodes, cnt.tris, cnt.wnodes, cnt.wtris };
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + p.red_off);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!count && k >= 4) break;
        const uint32_t s = k < 4 ? vals[k] : wave_sum(vals[k]);
        if (lane == 0) red[wave * 8 + k] = s;
    }
    __syncthreads();
    if (wave == 0 && lane < (count ? 8u : 4u)) {
        unsigned long long s = 0;
        for (int wv = 0; wv < trg::kWaves; ++wv) s += red[wv * 8 + lane];
        if (s) atomicAdd(&p.counters[(blockIdx.x % trg::kCounterSlots) * trg::kCounterWords + lane], s);
    }
}

// ---- head: raygen + bounces [0, K) of the chunk's frames; radiance so far -> radbuf; live paths -> the wavefront's queue segment ----
template <bool COUNT>
__global__ __launch_bounds__(trg::kBlock, TRG_EXP_WAVES) void render_head_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SceneView sc = scene_view<true>(p.sc, smem);
    sc.tex = p.tex;
    LdsStackT<trg::kBlock, false> stk;
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t bx, by;
    if (!block_tile(p, blockIdx.x, bx, by)) {   // a padding slot of the XCD-aware order: its wavefronts queue nothing
        if (lane_id() == 0u) p.tail_count[blockIdx.x * (uint32_t)trg::kWaves + wave] = 0u;
        return;
    }
    const uint32_t x0 = bx * trg::kTileW + (wave % (trg::kTileW / 8)) * 8;
    const uint32_t y0 = p.row0 + by * trg::kTileH + (wave / (trg::kTileW / 8)) * 8;   // row of the accumulation buffer
    const uint32_t y0i = image_row(p, y0);                                             // row of the image (interleaved bands)
    bool valid;
    uint32_t offset = 0u;
    {
        const uint32_t lane = lane_id();
        const uint32_t x = x0 + (lane & 7), y = y0 + (lane >> 3), yi = y0i + (lane >> 3);
        valid = (x < p.u.width) && (y < p.row0 + p.rows) && (yi < p.u.height);
        if (valid) offset = p.offsets[yi * p.u.width + x];
    }
    PathCounters pc; pc.primary = 0; pc.bounce = 0; pc.shadow = 0; pc.shaded = 0;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);
    constexpr bool TAB = !TRG_STRICT && TRG_HALTON_TABLES;
    typedef const __attribute__((address_space(4))) trg_uniforms cu_t;
    cu_t *up = (cu_t *)__builtin_amdgcn_kernarg_segment_ptr();
    v4f *queue = reinterpret_cast<v4f *>(p.tail_queue) + (size_t)(blockIdx.x * (uint32_t)trg::kWaves + wave) * p.tail_cap * 3u;
    v4f *radbuf = reinterpret_cast<v4f *>(p.tail_radbuf);
    uint32_t fill = 0;   // wave-uniform: entries this wavefront has appended to its segment
    for (uint32_t fl = 0; fl < p.spp; ++fl) {
        const uint32_t f = p.frame_begin + fl;
        const uint32_t lane_f = lane_id_opaque();
        const uint32_t xf = x0 + (lane_f & 7), yf = y0 + (lane_f >> 3);
        asm volatile("" : "+s"(up));
        V3 o, d;
        raygen<TAB>(*(const trg_uniforms *)up, xf, y0i + (lane_f >> 3), offset + f, o, d, sc.htab);
        V3 thr = mk(1.0f, 1.0f, 1.0f), rad = mk(0.0f, 0.0f, 0.0f);
        bool active = valid, primary_ray = true;
        pc.primary += wave_count(active);
        path_segment_lds<COUNT>(p, sc, stk, offset, f, 0u, p.tail_k, o, d, thr, rad, active, primary_ray, light_color, pc, cnt);
        const uint32_t pl = (yf - p.row0) * p.u.width + xf;   // pixel in the band
        if (valid) {
            v4f r4; r4.x = rad.x; r4.y = rad.y; r4.z = rad.z; r4.w = 0.0f;
            radbuf[(size_t)fl * p.tail_band_pixels + pl] = r4;
        }
        // the paths that go on: ballot / mbcnt prefix into this wavefront's own segment (a path that reaches here always has its
        // secondary mask: the only way to keep the primary ray is the invalid material, which the tail handles through the flag)
        const uint64_t m = __ballot(active);
        if (m != 0ull) {
            if (active) {
                v4f *e = queue + (size_t)(fill + mbcnt64(m)) * 3u;
                v4f e0, e1, e2;
                e0.x = o.x; e0.y = o.y; e0.z = o.z; e0.w = __uint_as_float(pl);
                e1.x = d.x; e1.y = d.y; e1.z = d.z; e1.w = __uint_as_float(fl | (primary_ray ? 0x80000000u : 0u));
                e2.x = thr.x; e2.y = thr.y; e2.z = thr.z; e2.w = 0.0f;
                e[0] = e0; e[1] = e1; e[2] = e2;
            }
            fill += (uint32_t)__popcll(m);
        }
    }
    const uint32_t lane = lane_id_opaque();
    if (lane == 0) p.tail_count[blockIdx.x * (uint32_t)trg::kWaves + wave] = fill;
    flush_counters(p, smem, wave, lane, pc, cnt, COUNT);
}

// ---- tail: bounces [tail_k, tail_k_end) for the queued paths, 64 consecutive entries of a segment per wavefront; when it stops
//      before the last bounce the survivors are compacted again into the wavefront's segment of the output queue ----
// (Round 3, measured and not kept: walking the shadow ray of bounce b and the nearest-hit ray of bounce b + 1 as a pair in one loop per
// lane, as the HBM kernels do.  Both rays are incoherent in a tail launch, so the wavefront would pay max(len_s + len_n) instead of
// max(len_s) + max(len_n) -- but the pair loop needs the any-hit flag per lane and 72 VGPRs: C3 at 64 spp 15.65 -> 17.7 ms per step at
// 7 waves/SIMD, 17.7 with spills at 8, 18.4 at 6.)
template <bool COUNT>
__global__ __launch_bounds__(trg::kBlock, TRG_EXP_WAVES) void render_tail_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SceneView sc = scene_view<true>(p.sc, smem);
    sc.tex = p.tex;
    LdsStackT<trg::kBlock, false> stk;
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t W = blockIdx.x * (uint32_t)trg::kWaves + wave;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.tail_count[W]);
    const v4f *queue = reinterpret_cast<const v4f *>(p.tail_queue) + (size_t)W * p.tail_cap * 3u;
    v4f *queue_out = reinterpret_cast<v4f *>(p.tail_queue_out) + (size_t)W * p.tail_cap * 3u;
    const bool requeue = p.tail_k_end < p.bounces;   // wave-uniform
    uint32_t fill = 0;
    v4f *radbuf = reinterpret_cast<v4f *>(p.tail_radbuf);
    PathCounters pc; pc.primary = 0; pc.bounce = 0; pc.shadow = 0; pc.shaded = 0;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);
    for (uint32_t base = 0; base < n; base += 64u) {   // wave-uniform trip count
        const uint32_t i = base + lane_id_opaque();
        bool active = i < n;
        V3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f), thr = mk(0.0f, 0.0f, 0.0f), rad = mk(0.0f, 0.0f, 0.0f);
        uint32_t pl = 0u, fl = 0u, offset = 0u;
        bool primary_ray = false;
        if (active) {
            const v4f e0 = queue[(size_t)i * 3u], e1 = queue[(size_t)i * 3u + 1u], e2 = queue[(size_t)i * 3u + 2u];
            o = mk(e0.x, e0.y, e0.z); d = mk(e1.x, e1.y, e1.z); thr = mk(e2.x, e2.y, e2.z);
            pl = __float_as_uint(e0.w);
            const uint32_t fbits = __float_as_uint(e1.w);
            fl = fbits & 0x7FFFFFFFu; primary_ray = (fbits >> 31) != 0u;
            const v4f r4 = radbuf[(size_t)fl * p.tail_band_pixels + pl];
            rad = mk(r4.x, r4.y, r4.z);
            if (p.il_n <= 1u) {
                offset = p.offsets[p.row0 * p.u.width + pl];
            } else {   // interleaved bands: `pl` counts pixels of the compact band; the Halton offsets are indexed by image pixel
                const uint32_t l = pl / p.u.width;
                offset = p.offsets[image_row(p, p.row0 + l) * p.u.width + (pl - l * p.u.width)];
            }
        }
        // `frame` must be wave-uniform for path_segment_lds (Halton index = offset + frame): the entries of one round may belong
        // to different frames, so the per-lane frame rides in `offset` and the uniform part is zero
        path_segment_lds<COUNT>(p, sc, stk, offset + p.frame_begin + fl, 0u, p.tail_k, p.tail_k_end, o, d, thr, rad, active, primary_ray, light_color, pc, cnt);
        if (i < n) {
            v4f r4; r4.x = rad.x; r4.y = rad.y; r4.z = rad.z; r4.w = 0.0f;
            radbuf[(size_t)fl * p.tail_band_pixels + pl] = r4;
        }
        if (requeue) {
            const uint64_t m = __ballot(active);
            if (m != 0ull) {
                if (active) {
                    v4f *e = queue_out + (size_t)(fill + mbcnt64(m)) * 3u;
                    v4f e0, e1, e2;
                    e0.x = o.x; e0.y = o.y; e0.z = o.z; e0.w = __uint_as_float(pl);
                    e1.x = d.x; e1.y = d.y; e1.z = d.z; e1.w = __uint_as_float(fl | (primary_ray ? 0x80000000u : 0u));
                    e2.x = thr.x; e2.y = thr.y; e2.z = thr.z; e2.w = 0.0f;
                    e[0] = e0; e[1] = e1; e[2] = e2;
                }
                fill += (uint32_t)__popcll(m);
            }
        }
    }
    if (requeue && lane_id_opaque() == 0) p.tail_count_out[W] = fill;
    flush_counters(p, smem, wave, lane_id_opaque(), pc, cnt, COUNT);
}

#if TRG_EXPERIMENTS
#include "../../experiments/trg_rtail.inc.h"   // render_rtail_kernel: tail launches with in-wave refill (TRG_OPT_TAIL_REFILL; measured slower: profiles/r05/c3_tail_refill_experiment.md)
#endif

// ---- sort a tile's queued paths by a coherence key (TRG_OPT_TAIL_SORT; round 4, DESIGN section 6: the verdict's "direction-octant sub-queues") ----
// A tail wavefront takes 64 CONSECUTIVE entries of its segment: paths of neighbouring pixels at bounce K, whose rays point anywhere.  This
// kernel re-orders the entries of a workgroup's four segments (one tile's paths of the chunk) by key -- the octant of the direction
// (3 bits) and, for mode >= 2, the cell of the origin in a 2^3 / 4^3 grid over the scene box -- so that the 64 lanes of a tail wavefront
// start from the same part of the room in the same general direction: same sign-ordered node addresses, similar walks.  The four
// segments keep their lengths (segment w takes the next count[w] entries of the sorted sequence), so the tail kernel is unchanged; order
// inside a key is whatever the LDS atomics give -- every path's arithmetic is its own, the image does not depend on it.
constexpr uint32_t kTailSortMaxKeys = 512u, kTailSortMaxEntries = 4u * 1024u;
TRG_DEV uint32_t tail_sort_key(const v4f e0, const v4f e1, uint32_t mode, const float *lo, const float *inv) {
    const uint32_t oct = (__float_as_uint(e1.x) >> 31) | ((__float_as_uint(e1.y) >> 31) << 1) | ((__float_as_uint(e1.z) >> 31) << 2);
    if (mode <= 1u) return oct;
    const uint32_t g = mode == 2u ? 2u : 4u;   // cells per axis
    const float fg = (float)g;
    const uint32_t cx = (uint32_t)fminf(fmaxf((e0.x - lo[0]) * inv[0] * fg, 0.0f), fg - 1.0f);
    const uint32_t cy = (uint32_t)fminf(fmaxf((e0.y - lo[1]) * inv[1] * fg, 0.0f), fg - 1.0f);
    const uint32_t cz = (uint32_t)fminf(fmaxf((e0.z - lo[2]) * inv[2] * fg, 0.0f), fg - 1.0f);
    return (((cz * g + cy) * g + cx) << 3) | oct;
}
__global__ __launch_bounds__(256) void tail_sort_kernel(const trg::RenderParams p, uint32_t mode, float lox, float loy, float loz, float ivx, float ivy, float ivz) {
    __shared__ unsigned short keys[kTailSortMaxEntries];
    __shared__ uint32_t hist[kTailSortMaxKeys];
    __shared__ uint32_t scan[kTailSortMaxKeys];
    __shared__ uint32_t segb[5];
    const uint32_t W0 = blockIdx.x * (uint32_t)trg::kWaves;
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (uint32_t w = 0; w < (uint32_t)trg::kWaves; ++w) { segb[w] = a; a += p.tail_count[W0 + w]; }
        segb[trg::kWaves] = a;
    }
    for (uint32_t k = threadIdx.x; k < kTailSortMaxKeys; k += 256u) hist[k] = 0u;
    __syncthreads();
    const uint32_t T = segb[trg::kWaves];
    if (T == 0u) return;
    const float lo[3] = { lox, loy, loz }, inv[3] = { ivx, ivy, ivz };
    const v4f *qin = reinterpret_cast<const v4f *>(p.tail_queue);
    v4f *qout = reinterpret_cast<v4f *>(p.tail_queue_out);
    // entry i of the tile's sequence -> (segment, index in it)
    auto locate = [&](uint32_t i, uint32_t &w, uint32_t &k) {
        w = (i >= segb[1]) + (i >= segb[2]) + (i >= segb[3]);
        k = i - segb[w];
    };
    for (uint32_t i = threadIdx.x; i < T; i += 256u) {
        uint32_t w, k;
        locate(i, w, k);
        const v4f *e = qin + ((size_t)(W0 + w) * p.tail_cap + k) * 3u;
        const uint32_t key = tail_sort_key(e[0], e[1], mode, lo, inv);
        keys[i] = (unsigned short)key;
        atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    // exclusive prefix over the keys (Hillis-Steele on 512 words, two per thread)
    for (uint32_t k = threadIdx.x; k < kTailSortMaxKeys; k += 256u) scan[k] = hist[k];
    __syncthreads();
    for (uint32_t d = 1u; d < kTailSortMaxKeys; d <<= 1) {
        uint32_t v[2];
        for (uint32_t j = 0; j < 2u; ++j) { const uint32_t k = threadIdx.x + j * 256u; v[j] = k >= d ? scan[k - d] : 0u; }
        __syncthreads();
        for (uint32_t j = 0; j < 2u; ++j) scan[threadIdx.x + j * 256u] += v[j];
        __syncthreads();
    }
    for (uint32_t k = threadIdx.x; k < kTailSortMaxKeys; k += 256u) hist[k] = scan[k] - hist[k];   // first position of key k
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < T; i += 256u) {
        uint32_t w, k;
        locate(i, w, k);
        const v4f *e = qin + ((size_t)(W0 + w) * p.tail_cap + k) * 3u;
        const v4f e0 = e[0], e1 = e[1], e2 = e[2];
        const uint32_t pos = atomicAdd(&hist[keys[i]], 1u);
        uint32_t w2, k2;
        locate(pos, w2, k2);
        v4f *o = qout + ((size_t)(W0 + w2) * p.tail_cap + k2) * 3u;
        o[0] = e0; o[1] = e1; o[2] = e2;
    }
}

// ---- fold the chunk's frames into the running average, in frame order (Accumulate.metal:19-39) ----
__global__ __launch_bounds__(256) void tail_accumulate_kernel(const trg::RenderParams p) {
    const uint32_t pl = blockIdx.x * 256u + threadIdx.x;
    if (pl >= p.tail_band_pixels) return;
    v4f *accum = reinterpret_cast<v4f *>(p.accum);
    const v4f *radbuf = reinterpret_cast<const v4f *>(p.tail_radbuf);
    const uint32_t pix = p.row0 * p.u.width + pl;
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    if (p.frame_begin > 0) { const v4f a = accum[pix]; acc = mk(a.x, a.y, a.z); }
    for (uint32_t fl = 0; fl < p.spp; ++fl) {
        const v4f r4 = radbuf[(size_t)fl * p.tail_band_pixels + pl];
        const V3 rad = mk(r4.x, r4.y, r4.z);
        const uint32_t f = p.frame_begin + fl;
        if (f == 0) {
            acc = rad;
        } else {
            const V3 prev = acc * (float)f;
            const V3 c = rad + prev;
            const float f1 = (float)(f + 1u);
            acc = mk(c.x / f1, c.y / f1, c.z / f1);
        }
    }
    v4f outv; outv.x = acc.x; outv.y = acc.y; outv.z = acc.z; outv.w = 1.0f;
    accum[pix] = outv;
}
