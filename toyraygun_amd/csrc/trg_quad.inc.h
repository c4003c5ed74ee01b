// trg_quad.inc.h -- quad traversal of the quantised 4-wide tree (scenes in HBM); included by trg_device.h inside namespace trgdev.
//
// Why: with one ray per lane the HBM megakernel is bound by the texture addresser and by VALU issue at the same time, at a quarter
// of the lanes doing useful work (profiles/r02: 441 M vector loads per C4 launch at 23.5 TA-busy cycles each = 65-70 % of the
// launch; 16.3 of 64 lanes active per VALU instruction).  A 16-byte load occupies the TA for 16 cycles per wavefront whatever the
// lanes address, a 64-byte node is four of them, and the four child boxes are tested one after the other by the one lane that owns
// the ray while the lanes whose rays are finished wait for it.
//
// Here FOUR lanes share one ray.  The 64 lanes of a wavefront are 16 quads; a quad fetches a node with ONE load instruction (lane k
// of the quad loads the k-th 16 bytes), the header words travel between the four lanes as DPP operands, and every lane tests ONE
// child box.  The four (distance, child) results are ordered inside the quad with two DPP min steps and three DPP compares, the
// hit children go onto the quad's stack in LDS farthest first, and the next node is read back from the top.  A leaf of up to four
// triangles is tested one triangle per lane.  The wavefront's rays (up to two per lane: the shadow ray and the continuation ray a
// shading event produces) are handed to quads as they become free, so no quad waits for the slowest ray of the wavefront and the
// lanes stay busy until fewer than 16 rays are left: the same rays, the same arithmetic per ray (tri_test and the slab products
// are the functions of the one-ray-per-lane path), a quarter of the load instructions and well under half of the VALU
// instructions per node.
//
// The intersection contract (nearest hit, ties to the lower primitive index; any-hit: any accepted triangle) does not depend on the
// order in which nodes and triangles are visited, so the results are bit-identical to the other schedules' -- and to the oracle's
// in the strict build.
#pragma once

// ---- DPP / cross-lane helpers (all four lanes of a quad are always in the same control flow) ----
template <int CTRL>
TRG_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
TRG_DEV float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
#define TRG_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
constexpr int kQpBcast0 = TRG_QP(0, 0, 0, 0), kQpBcast1 = TRG_QP(1, 1, 1, 1), kQpBcast2 = TRG_QP(2, 2, 2, 2), kQpBcast3 = TRG_QP(3, 3, 3, 3);
constexpr int kQpXor1 = TRG_QP(1, 0, 3, 2), kQpXor2 = TRG_QP(2, 3, 0, 1);
constexpr int kQpRot1 = TRG_QP(1, 2, 3, 0), kQpRot2 = TRG_QP(2, 3, 0, 1), kQpRot3 = TRG_QP(3, 0, 1, 2);

TRG_DEV int bperm_i(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
TRG_DEV float bperm_f(int src_lane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }

// LDS a quad tracer uses: the quads' stacks (kQuadStride ints each, level 0 = the kNodeDone sentinel) and, per thread, the result
// slots the quads write for the owner lanes: (t | -1, prim, u, v) of the nearest-hit ray and one word for the shadow ray.
struct QuadLds {
    uint32_t stack;   // LDS byte address of level 0 of this lane's quad
    uint32_t res;     // LDS byte address of result slot 0 of this wavefront's lane 0 (slot of lane L: + L * 4, component c: + c * BLOCK * 4)
};
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
TRG_DEV void lds_st(uint32_t addr, int v) { *(lds_int_t *)(uintptr_t)addr = v; }
TRG_DEV int lds_ld(uint32_t addr) { return *(lds_int_t *)(uintptr_t)addr; }

// per-quad traversal state (identical in the four lanes of a quad, except where noted)
struct QuadRay {
    V3 o, d;
    float idx, idy, idz, oix, oiy, oiz;
    float tmax, best;
    uint32_t rmask;
    int prim; float u, v;   // best hit so far (t = best when found)
    bool found, any;
    bool negx, negy, negz;
    int node, sp;           // sp: index of the next free stack entry (1 = empty: level 0 holds the sentinel)
};

TRG_DEV void quad_begin(QuadRay &q, V3 o, V3 d, float tmax, uint32_t rmask, bool any) {
    const float dx = clamp_away_from_zero(d.x), dy = clamp_away_from_zero(d.y), dz = clamp_away_from_zero(d.z);
    q.o = o; q.d = d;
    q.idx = rcp_fast(dx); q.idy = rcp_fast(dy); q.idz = rcp_fast(dz);
    q.oix = o.x * q.idx; q.oiy = o.y * q.idy; q.oiz = o.z * q.idz;
    q.tmax = tmax; q.best = tmax; q.rmask = rmask;
    q.prim = -1; q.u = 0.0f; q.v = 0.0f; q.found = false; q.any = any;
    q.negx = (__float_as_uint(dx) >> 31) != 0u; q.negy = (__float_as_uint(dy) >> 31) != 0u; q.negz = (__float_as_uint(dz) >> 31) != 0u;
    q.node = 0; q.sp = 1;
}

// One unit of work for every busy quad off ONE load instruction: an inner node (lane k loads the k-th 16 bytes and tests child k) or
// one triangle of the current leaf (lanes 0..2 load its three 16-byte rows; all four lanes run the same test).  The twelve header /
// triangle words reach the four lanes as DPP broadcasts either way.  ~node = (first << 3) | (remaining - 1), consumed by advancing it.
template <bool COUNT>
TRG_DEV void quad_step(const SceneView &sc, QuadRay &q, uint32_t stack, uint32_t k, Counters &cnt) {
    const bool inner = q.node >= 0;
    const uint32_t code = (uint32_t)~q.node;
    const uint32_t first = code >> 3, left = code & 7u;
    const v4f *ptr = inner ? sc.nodes + ((size_t)q.node * 4 + k) : sc.tris + ((size_t)first * 3 + (k < 2u ? k : 2u));
    const v4f c = *ptr;
    const float a0 = dpp_f<kQpBcast0>(c.x), a1 = dpp_f<kQpBcast0>(c.y), a2 = dpp_f<kQpBcast0>(c.z), a3 = dpp_f<kQpBcast0>(c.w);
    const float b0 = dpp_f<kQpBcast1>(c.x), b1 = dpp_f<kQpBcast1>(c.y), b2 = dpp_f<kQpBcast1>(c.z), b3 = dpp_f<kQpBcast1>(c.w);
    const float c0 = dpp_f<kQpBcast2>(c.x), c1 = dpp_f<kQpBcast2>(c.y), c2 = dpp_f<kQpBcast2>(c.z), c3 = dpp_f<kQpBcast2>(c.w);
    if (inner) {
        if (COUNT) { if (k == 0u) cnt.nodes += 2; if (mbcnt64(__ballot(1)) == 0) cnt.wnodes++; }
        // q4node.h: row 0 = origin, x scale; row 1 = lo.x hi.x lo.y hi.y (a byte per child); row 2 = lo.z hi.z, y and z scale; row 3 = children
        const int ch0 = dpp_i<kQpBcast3>(__float_as_int(c.x)), ch1 = dpp_i<kQpBcast3>(__float_as_int(c.y));
        const int ch2 = dpp_i<kQpBcast3>(__float_as_int(c.z)), ch3 = dpp_i<kQpBcast3>(__float_as_int(c.w));
        const int child = k == 0u ? ch0 : (k == 1u ? ch1 : (k == 2u ? ch2 : ch3));
        const float ax = a3 * q.idx, ay = c2 * q.idy, az = c3 * q.idz;
        const float bx = a0 * q.idx - q.oix, by = a1 * q.idy - q.oiy, bz = a2 * q.idz - q.oiz;
        const uint32_t lox = __float_as_uint(b0), hix = __float_as_uint(b1), loy = __float_as_uint(b2), hiy = __float_as_uint(b3);
        const uint32_t loz = __float_as_uint(c0), hiz = __float_as_uint(c1);
        const uint32_t sh = k * 8u;
        const uint32_t nx = q.negx ? hix : lox, fx = q.negx ? lox : hix;
        const uint32_t ny = q.negy ? hiy : loy, fy = q.negy ? loy : hiy;
        const uint32_t nz = q.negz ? hiz : loz, fz = q.negz ? loz : hiz;
        const float tnx = (float)((nx >> sh) & 255u) * ax + bx, tfx = (float)((fx >> sh) & 255u) * ax + bx;
        const float tny = (float)((ny >> sh) & 255u) * ay + by, tfy = (float)((fy >> sh) & 255u) * ay + by;
        const float tnz = (float)((nz >> sh) & 255u) * az + bz, tfz = (float)((fz >> sh) & 255u) * az + bz;
        const float tmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tmax = fminf(fminf(tfx, tfy), fminf(tfz, q.best));
        // order the four results: key = entry distance with the lane number in the two lowest mantissa bits (distances are >= 0, so
        // the bit patterns order like the floats; the two bits only break ties), misses above every hit.  Everything below is
        // derived from the keys alone, so the four lanes always agree.  (An unused slot decodes to the inverted box: never entered.)
        const uint32_t key = (tmin <= tmax) ? ((__float_as_uint(tmin) & 0x7FFFFFFCu) | k) : (0x7F800000u | k);
        const bool hit = key < 0x7F800000u;   // entered, and not at t = +inf
        const uint32_t k1 = (uint32_t)dpp_i<kQpRot1>((int)key), k2 = (uint32_t)dpp_i<kQpRot2>((int)key), k3 = (uint32_t)dpp_i<kQpRot3>((int)key);
        const uint32_t rank = (k1 < key ? 1u : 0u) + (k2 < key ? 1u : 0u) + (k3 < key ? 1u : 0u);
        const uint32_t h = (hit ? 1u : 0u) + (k1 < 0x7F800000u ? 1u : 0u) + (k2 < 0x7F800000u ? 1u : 0u) + (k3 < 0x7F800000u ? 1u : 0u);
        const uint32_t kmin = min(min(key, k1), min(k2, k3));
        const uint32_t n = kmin & 3u;   // the lane of the nearest entered child (when h > 0)
        const int nearest = n == 0u ? ch0 : (n == 1u ? ch1 : (n == 2u ? ch2 : ch3));
        // the farther entered children go onto the stack, farthest first; the nearest is the next node; with no child entered the
        // pending entry below is popped (the sentinel when nothing is pending)
        if (hit && rank != 0u) lds_st(stack + ((uint32_t)q.sp + h - 1u - rank) * 4u, child);
        int sp = q.sp + (h != 0u ? (int)h - 1 : 0);
        int next = nearest;
        if (h == 0u) { sp -= 1; next = lds_ld(stack + (uint32_t)sp * 4u); }
        q.node = next; q.sp = sp;
    } else {
        v4f ta, tb, tc;
        ta.x = a0; ta.y = a1; ta.z = a2; ta.w = a3; tb.x = b0; tb.y = b1; tb.z = b2; tb.w = b3; tc.x = c0; tc.y = c1; tc.z = c2; tc.w = c3;
        const bool masked_in = (((uint32_t)__float_as_int(tb.w)) & q.rmask) != 0u;
        if (COUNT) { if (masked_in && k == 0u) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
        float t, u, v;
        const bool ok = tri_test(ta, tb, tc, q.o, q.d, q.tmax, t, u, v) && masked_in;
        const int prim = __float_as_int(ta.w);
        const bool take = ok && (q.any || !q.found || t < q.best || (t == q.best && prim < q.prim));
        q.found = q.found || ok;
        q.best = (take && !q.any) ? t : q.best;
        q.prim = take ? prim : q.prim;
        q.u = take ? u : q.u;
        q.v = take ? v : q.v;
        const bool stop = q.any && ok;
        const bool more = left != 0u;
        const bool do_pop = !stop && !more;
        const int sp = q.sp - (do_pop ? 1 : 0);
        int next = stop ? kNodeDone : ~(int)(((first + 1u) << 3) | (left - 1u));
        if (do_pop) next = lds_ld(stack + (uint32_t)sp * 4u);
        q.node = next; q.sp = sp;
    }
}

// The wavefront's rays -- per lane an optional any-hit ray A (mask 1) and an optional nearest-hit ray B, both from `org` -- traced by
// its 16 quads.  Must be called by all 64 lanes in uniform control flow.  `refill_min`: quads are handed new rays once this many
// lanes are idle (or nothing is running).
template <bool COUNT, int BLOCK>
TRG_DEV void traverse_quads(const SceneView &sc, const QuadLds ql, V3 org, bool has_a, V3 adir, float amax, bool has_b, V3 bdir, uint32_t bmask,
                            bool &occluded, Hit &bhit, bool &bfound, Counters &cnt) {
    const uint32_t lane = mbcnt64(~0ull);
    const uint32_t k = lane & 3u;
    const uint32_t my_res = ql.res + lane * 4u;
    uint64_t pend_a = __ballot(has_a), pend_b = __ballot(has_b);
    int job = -1;   // owner lane | kind << 6 (kind 1 = nearest-hit ray B), -1 = this quad is idle
    QuadRay q;
    quad_begin(q, mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 1.0f), 0.0f, 0u, true);
    q.node = kNodeDone;
#ifdef TRG_QUAD_GUARD
    uint32_t guard = 0;
#endif
    for (;;) {
#ifdef TRG_QUAD_GUARD
        if (++guard > (1u << 22)) break;   // bring-up only: a malformed stack must not hang the GPU
#endif
        const uint64_t idle = __ballot(job < 0);
        const bool pending = (pend_a | pend_b) != 0ull;
        if (!pending && idle == ~0ull) break;
        if (pending && __popcll(idle) >= 16) {
            // hand the next rays to idle quads: the long ones (nearest-hit) first
            uint64_t leaders = idle & 0x1111111111111111ull;
            const int before = job;
            while (leaders != 0ull && (pend_a | pend_b) != 0ull) {
                const int ql0 = __builtin_ctzll(leaders);
                leaders &= leaders - 1ull;
                int j;
                if (pend_b != 0ull) { j = __builtin_ctzll(pend_b) | 64; pend_b &= pend_b - 1ull; }
                else { j = __builtin_ctzll(pend_a); pend_a &= pend_a - 1ull; }
                job = (int)(lane & ~3u) == ql0 ? j : job;
            }
            // the rays come out of the owner lanes' registers (every lane executes the permutes: a disabled lane would read as 0)
            const int owner = job & 63;
            const bool kind_b = (job & 64) != 0;
            const V3 o = mk(bperm_f(owner, org.x), bperm_f(owner, org.y), bperm_f(owner, org.z));
            const V3 da = mk(bperm_f(owner, adir.x), bperm_f(owner, adir.y), bperm_f(owner, adir.z));
            const V3 db = mk(bperm_f(owner, bdir.x), bperm_f(owner, bdir.y), bperm_f(owner, bdir.z));
            const float ta = bperm_f(owner, amax);
            const uint32_t mb = (uint32_t)bperm_i(owner, (int)bmask);
            if (job != before) quad_begin(q, o, kind_b ? db : da, kind_b ? INFINITY : ta, kind_b ? mb : 1u, !kind_b);
        }
        if (job >= 0) {
            quad_step<COUNT>(sc, q, ql.stack, k, cnt);
            if (q.node == kNodeDone) {
                if (k == 0u) {
                    const uint32_t slot = ql.res + (uint32_t)(job & 63) * 4u;
                    if (job & 64) {
                        lds_st(slot, __float_as_int(q.found ? q.best : -1.0f));
                        lds_st(slot + BLOCK * 4u, q.prim);
                        lds_st(slot + 2u * BLOCK * 4u, __float_as_int(q.u));
                        lds_st(slot + 3u * BLOCK * 4u, __float_as_int(q.v));
                    } else {
                        lds_st(slot + 4u * BLOCK * 4u, q.found ? 1 : 0);
                    }
                }
                job = -1;
            }
        }
    }
    occluded = false; bfound = false;
    bhit.t = -1.0f; bhit.prim = -1; bhit.u = 0.0f; bhit.v = 0.0f;
    if (has_a) occluded = lds_ld(my_res + 4u * BLOCK * 4u) != 0;
    if (has_b) {
        bhit.t = __int_as_float(lds_ld(my_res));
        bhit.prim = lds_ld(my_res + BLOCK * 4u);
        bhit.u = __int_as_float(lds_ld(my_res + 2u * BLOCK * 4u));
        bhit.v = __int_as_float(lds_ld(my_res + 3u * BLOCK * 4u));
        bfound = bhit.prim >= 0;
    }
}
