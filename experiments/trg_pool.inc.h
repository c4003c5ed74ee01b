// experiments/trg_pool.inc.h -- EXPERIMENT, not in the shipped library: the workgroup PATH POOL schedule (TRG_KERNEL_POOL).
// Included by toyraygun_amd/csrc/trg_kernels.hip inside namespace trgk_{fast,strict} when TRG_EXPERIMENTS=1 (experiments/build.py).
// Built in round 1, bit-exact against the oracle, and slower than the direct megakernel everywhere it was measured (C2 3.48 vs 2.18 ms, 8 bounces
// 7.87 vs 5.90 ms: 114-122 VGPRs and 54 KB of LDS per workgroup leave 3 waves per SIMD) -- NOTEBOOK.md, "Compaction".
#pragma once

// ---------------------------------------------------------------------------------------------
// Queue-draining tracer for the path-pool megakernel (render_pool_kernel).
//
// A workgroup keeps a pool of path slots in LDS; a compacted list names the rays to trace this phase
// (nearest-hit rays of the current bounce and any-hit shadow rays of the previous one, mixed).  Every
// lane of every wavefront is a worker: when it has no ray it takes the next list entry (one LDS atomic
// per wavefront per refill, slots handed out with a ballot / mbcnt prefix), traverses, writes the result
// to the slot and comes back for more.  Lanes whose rays end early therefore do not idle until the
// slowest lane of the wavefront is done -- the reason secondary rays ran at 23-46 % lane utilisation in
// the one-ray-per-lane kernel.  A refill is attempted only when >= kRefillMin lanes are idle (or all are).
//
// Slot layout (float4 arrays in LDS, P = slots per workgroup):
//   R0[s] = (origin.xyz, maxDistance)   R1[s] = (direction.xyz, bits mask)      nearest-hit ray
//   SH[s] = (shadow direction.xyz, shadow maxDistance; set to -1 by the tracer when occluded);
//           the shadow ray starts at R0[s].xyz (same origin as the next bounce ray); tracers never write R0/R1
//   H[s]  = (t or -1, bits primitiveIndex, u, v)                                 nearest-hit result
// List entry = slot | (kind << 15), kind 0 = nearest, 1 = shadow (any-hit, mask 1).
// ---------------------------------------------------------------------------------------------
#ifndef TRG_REFILL_MIN
#define TRG_REFILL_MIN 16
#endif
constexpr int kRefillMin = TRG_REFILL_MIN;

struct PoolView {
    v4f *R0, *R1, *SH, *H;
};

template <bool COUNT, int BLOCK, bool UNIFIED = false, typename STK>
TRG_DEV void trace_queue(const SceneView &sc, const PoolView pv, const unsigned short *list, uint32_t count,
                         uint32_t *head, STK stk, Counters &cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    bool busy = false, exhausted = false, any = false;
    uint32_t slot = 0;
    Trav tv;
    trav_begin(sc, tv, mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 1.0f), 0.0f, 0u, stk.first(), lds_node_base<UNIFIED>(sc), !UNIFIED);
    tv.node = kNodeDone;
    for (;;) {
        // ---- refill: idle lanes take the next list entries ----
        const uint64_t idle = __ballot(!busy);
        if (!exhausted && idle != 0ull) {
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const int leader = __ffsll((long long)idle) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(head, n_idle);
            base = (uint32_t)__shfl((int)base, leader, 64);
            if (base + n_idle >= count) exhausted = true;
            const uint32_t my = base + mbcnt64(idle);
            if (!busy && my < count) {
                const uint32_t e = list[my];
                slot = e & 0x7FFFu;
                any = (e >> 15) != 0u;
                const v4f r0 = pv.R0[slot];  // the shadow ray starts where the next ray starts
                const v4f r1 = any ? pv.SH[slot] : pv.R1[slot];
                trav_begin(sc, tv, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), any ? r1.w : r0.w,
                           any ? 1u : (uint32_t)__float_as_int(r1.w), stk.first(), lds_node_base<UNIFIED>(sc), !UNIFIED);
                busy = true;
            }
        }
        if (__ballot(busy) == 0ull) break;
        if (busy) {
            for (;;) {
                if (UNIFIED) {
                    trav_step_hbm<COUNT, BLOCK>(sc, tv, any, stk, cnt);
                } else {
                    while (tv.node >= 0) trav_node_step_signed<COUNT, BLOCK>(sc, tv, stk, cnt);
                    if (tv.node == kNodeDone) break;
                    if (trav_leaf_step<COUNT, BLOCK>(sc, tv, any, stk, cnt)) tv.node = kNodeDone;
                }
                if (tv.node == kNodeDone) break;
                // enough lanes of this wavefront have run dry: let them refill (state stays in registers)
                if (!exhausted && __popcll(__ballot(1)) <= 64 - kRefillMin) break;
            }
            if (tv.node == kNodeDone) {
                if (any) {
                    if (tv.found) { v4f sh = pv.SH[slot]; sh.w = -1.0f; pv.SH[slot] = sh; }
                } else {
                    v4f h;
                    Hit bh = trav_hit(tv);      // (a hit that is still a BOX gets its triangle and weights here: trg_device.h box_hit_resolve / lds_box_hit_resolve)
                    box_hit_resolve(sc, tv, bh); lds_box_hit_resolve(sc, tv, bh);
                    h.x = bh.t; h.y = __int_as_float(bh.prim); h.z = bh.u; h.w = bh.v;
                    pv.H[slot] = h;
                }
                busy = false;
            }
        }
    }
}

// wave-compacted append of `val` to an LDS list: one atomic per wavefront, slots by ballot prefix
TRG_DEV void list_append(bool pred, unsigned short val, unsigned short *list, uint32_t *counter) {
    const uint64_t m = __ballot(pred);
    if (m != 0ull) {
        const uint32_t lane = threadIdx.x & 63u;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (pred) list[base + mbcnt64(m)] = val;
    }
}


// ---------------------------------------------------------------------------------------------
// render_pool_kernel: the same path tracer, scheduled as a workgroup-local wavefront.
//
// A workgroup owns a 16x16 pixel tile and S consecutive frames at a time: a pool of P = 256*S path
// slots in LDS.  Thread t owns the S paths of its pixel (throughput, radiance, running average stay in
// ITS registers, so the per-pixel accumulation order is the reference's), but rays are traced by
// whichever lane is free: each bounce is one TRACE phase (trace_queue drains the compacted list of
// nearest-hit rays of this bounce + shadow rays of the previous bounce) and one SHADE phase (owners
// consume hit records, emit the next rays and append them to the next list with ballot/prefix
// compaction).  Dead paths simply stop appearing in the lists.  Arithmetic per path is identical to
// render_kernel: only the schedule differs (the strict build of both is bit-exact against the oracle).
// ---------------------------------------------------------------------------------------------
template <bool LDS_SCENE, bool COUNT, int S>
__global__ __launch_bounds__(trg::kBlock) void render_pool_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SceneView sc = scene_view<LDS_SCENE>(p.sc, smem);
    sc.tex = p.tex;
    LdsStackT<trg::kBlock, !LDS_SCENE> stk;  // HBM scenes may spill deep stack levels to global scratch
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);  // also writes the sentinel at level 0 of this thread's column
    constexpr int P = trg::kBlock * S;
    PoolView pv;
    pv.R0 = reinterpret_cast<v4f *>(smem + p.pool_off);
    pv.R1 = pv.R0 + P; pv.SH = pv.R1 + P; pv.H = pv.SH + P;
    unsigned short *lists = reinterpret_cast<unsigned short *>(pv.H + P);  // two lists of 2*P entries
    uint32_t *ctr = reinterpret_cast<uint32_t *>(lists + 4 * P);         // [0] head, [1] n(list 0), [2] n(list 1)

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t bx, by;
    if (!block_tile(p, blockIdx.x, bx, by)) return;
    const uint32_t x = bx * trg::kTileW + (wave % (trg::kTileW / 8)) * 8 + (lane & 7);
    const uint32_t y = p.row0 + by * trg::kTileH + (wave / (trg::kTileW / 8)) * 8 + (lane >> 3);
    const bool valid = (x < p.u.width) && (y < p.row0 + p.rows) && (y < p.u.height);
    const uint32_t pix = y * p.u.width + x;

    uint32_t n_primary = 0, n_bounce = 0, n_shadow = 0, n_shaded = 0;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
    const uint32_t offset = valid ? p.offsets[pix] : 0u;
    v4f *accum = reinterpret_cast<v4f *>(p.accum);
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    if (valid && p.frame_begin > 0) { const v4f a = accum[pix]; acc = mk(a.x, a.y, a.z); }
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);
    const uint32_t frame_end = p.frame_begin + p.spp;

    V3 thr[S], rad[S], scol[S];
    bool alive[S], shpend[S];

    for (uint32_t f0 = p.frame_begin; f0 < frame_end; f0 += S) {
        // ---------------- raygen: owners fill their slots, list 0 = every valid path ----------------
        if (threadIdx.x == 0) { ctr[0] = 0u; ctr[1] = 0u; ctr[2] = 0u; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const uint32_t slot = threadIdx.x + trg::kBlock * j;
            const bool pvalid = valid && (f0 + j < frame_end);
            thr[j] = mk(1.0f, 1.0f, 1.0f); rad[j] = mk(0.0f, 0.0f, 0.0f); scol[j] = mk(0.0f, 0.0f, 0.0f);
            alive[j] = pvalid; shpend[j] = false;
            if (pvalid) {
                V3 o, d;
                raygen(p.u, x, y, offset + f0 + j, o, d);
                v4f r0, r1;
                r0.x = o.x; r0.y = o.y; r0.z = o.z; r0.w = INFINITY;
                r1.x = d.x; r1.y = d.y; r1.z = d.z; r1.w = __int_as_float(3);  // RAY_MASK_PRIMARY
                pv.R0[slot] = r0; pv.R1[slot] = r1;
                n_primary++;
            }
            list_append(pvalid, (unsigned short)slot, lists, &ctr[1]);
        }
        __syncthreads();

        uint32_t cur = 0;
        for (uint32_t b = 0; b <= p.bounces; ++b) {
            const uint32_t n_cur = ctr[1 + cur];
            if (n_cur == 0u) break;  // workgroup-uniform: nothing left in flight
            // ---------------- TRACE ----------------
            trace_queue<COUNT, trg::kBlock, !LDS_SCENE>(sc, pv, lists + cur * 2 * P, n_cur, &ctr[0], stk, cnt);
            __syncthreads();
            if (threadIdx.x == 0) { ctr[0] = 0u; ctr[1 + cur] = 0u; }  // head, and this list becomes the next "next"
            // ---------------- SHADE ----------------
            unsigned short *next = lists + (cur ^ 1u) * 2 * P;
            uint32_t *n_next = &ctr[1 + (cur ^ 1u)];
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const uint32_t slot = threadIdx.x + trg::kBlock * j;
                if (shpend[j]) {  // shadowHit (Raytracing.metal:219-246) of the previous bounce
                    if (pv.SH[slot].w >= 0.0f) rad[j] = rad[j] + scol[j];
                    shpend[j] = false;
                }
                bool trace_next = false, trace_shadow = false;
                if (alive[j] && b < p.bounces) {
                    const v4f h = pv.H[slot];
                    if (!(h.x >= 0.0f)) {
                        alive[j] = false;  // Raytracing.metal:139-144
                    } else {
                        const int prim = __float_as_int(h.y);   // scene in HBM: the leaf record of the hit
                        const float hu = h.z, hv = h.w;
                        const Surf<!LDS_SCENE> sf = surf_fetch<!LDS_SCENE>(sc, prim);
                        const uint32_t mat = sf.mat;
                        if (mat == TRG_MATERIAL_DEFAULT) {
                            n_shaded++;
                            const v4f r0 = pv.R0[slot], r1 = pv.R1[slot];
                            const V3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z);
                            const V3 Pp = o + d * h.x;
                            const float cx = 1.0f - hu - hv, cy = hu;
                            V3 vcol, nraw;
                            surf_interp<!LDS_SCENE>(sc, sf, cx, cy, vcol, nraw);
                            const V3 nrm = normalize(nraw);
                            float r[4];
                            uint32_t hi = offset + f0 + j;
                            asm volatile("" : "+v"(hi));
                            const bool last = (b + 1u == p.bounces);  // workgroup-uniform
                            r[2] = 0.0f; r[3] = 0.0f;
                            if (last) halton2(hi, b, r); else halton4(hi, b, r);
                            const LightSample ls = sample_area_light(p.u, r[0], r[1], Pp, nrm);
                            thr[j] = thr[j] * vcol;
                            const V3 so = Pp + nrm * 1e-3f;
                            const float smax = ls.dist - 1e-3f;
                            scol[j] = ls.color * thr[j];
                            V3 nd = nrm;
                            if (!last) nd = align_hemisphere(sample_cosine_hemisphere(r[2], r[3]), nrm);
                            v4f q0, q1, sh;
                            q0.x = so.x; q0.y = so.y; q0.z = so.z; q0.w = INFINITY;
                            q1.x = nd.x; q1.y = nd.y; q1.z = nd.z; q1.w = __int_as_float(1);  // RAY_MASK_SECONDARY
                            sh.x = ls.dir.x; sh.y = ls.dir.y; sh.z = ls.dir.z; sh.w = smax;
                            pv.R0[slot] = q0; pv.R1[slot] = q1; pv.SH[slot] = sh;
                            trace_shadow = smax >= 0.0f;
                            trace_next = (b + 1u < p.bounces);
                            if (trace_shadow) { shpend[j] = true; n_shadow++; }
                            if (trace_next) n_bounce++;
                        } else if (mat == TRG_MATERIAL_EMISSIVE) {
                            rad[j] = light_color;
                            alive[j] = false;
                        } else {
                            rad[j] = mk(1.0f, 0.0f, 1.0f);  // ray left as is: traced again unchanged
                            trace_next = (b + 1u < p.bounces);
                            if (trace_next) n_bounce++;
                        }
                    }
                }
                list_append(trace_next, (unsigned short)slot, next, n_next);
                list_append(trace_shadow, (unsigned short)(slot | 0x8000u), next, n_next);
            }
            __syncthreads();
            cur ^= 1u;
        }
        // ---------------- accumulate (Accumulate.metal:19-39), frames in order ----------------
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const uint32_t f = f0 + j;
            if (f < frame_end) {
                if (f == 0) {
                    acc = rad[j];
                } else {
                    const V3 prev = acc * (float)f;
                    const V3 c = rad[j] + prev;
                    const float f1 = (float)(f + 1u);
                    acc = mk(c.x / f1, c.y / f1, c.z / f1);
                }
            }
        }
    }
    if (valid) {
        v4f outv; outv.x = acc.x; outv.y = acc.y; outv.z = acc.z; outv.w = 1.0f;
        accum[pix] = outv;
    }

    uint32_t vals[8] = { n_primary, n_bounce, n_shadow, n_shaded, cnt.nodes, cnt.tris, cnt.wnodes, cnt.wtris };
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + p.red_off);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!COUNT && k >= 4) break;
        const uint32_t s = wave_sum(vals[k]);
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

