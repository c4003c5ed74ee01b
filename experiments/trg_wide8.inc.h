// experiments/trg_wide8.inc.h -- EXPERIMENT, not in the shipped library (-DTRG_WIDE8=1: experiments/build.py builds
// experiments/lib/libtoyraygun_hip_w8.so): scenes in HBM traversed through COMPRESSED 8-WIDE nodes with octant-ordered slots (q8node.h; round-4
// verdict, item 3a).  Bit-exact in the strict build; 12 % slower than the quantised 4-wide tree on C4, 22 % on the 10.6 M-triangle scene, with the
// counters in profiles/r05/c4_wide8_experiment.md.  Included by toyraygun_amd/csrc/trg_device.h inside namespace trgdev.
#pragma once

// One unit of work per lane per iteration on the COMPRESSED 8-WIDE tree (q8node.h): the next leaf record of the leaf group in hand, or --
// none pending -- the nearest not yet entered inner child of the node group in hand: its 80-byte node = five 16-byte loads, eight slab
// tests whose outcomes are shifted into one byte (v_sub + v_alignbit per child: the sign of tmax - tmin), no sorting network: the hit inner
// slots are put in priority order by XOR-ing their bit positions with the ray's octant, the rest of the old group goes to the stack as
// ONE entry of two words.  Leaf slots that were hit are tested straight away, before any inner child (Ylitie et al. 2017, section 4).
template <bool COUNT, int BLOCK, typename STK>
TRG_DEV void trav_step_wide8(const SceneView &sc, Trav &tv, bool any, STK stk, Counters &cnt) {
    const bool tri = (tv.t_bits & 0xFFu) != 0u;
    const uint32_t octinv = ((__float_as_uint(tv.idx) >> 31) | ((__float_as_uint(tv.idy) >> 31) << 1) | ((__float_as_uint(tv.idz) >> 31) << 2)) ^ 7u;
    uint32_t off, rec = 0u;
    if (tri) {
        const uint32_t sl = (uint32_t)__builtin_ctz(tv.t_bits & 0xFFu);
        const uint32_t below = (tv.t_bits >> 8) & ((1u << sl) - 1u);
        rec = tv.t_base + 2u * (uint32_t)__popc(below) + ((tv.t_bits >> 16) & 1u);
        off = rec * (uint32_t)(kRecV4 * 16) + sc.rec_delta;
    } else {
        const uint32_t p = 31u - (uint32_t)__builtin_clz(tv.g_bits & 0xFFu);   // (a lane that is here has a pending inner child: trav_begin / the pop below)
        const uint32_t sl = p ^ octinv;
        const uint32_t below = (tv.g_bits >> 8) & ((1u << sl) - 1u);
        off = (tv.g_base + (uint32_t)__popc(below)) * 80u;
        tv.g_bits &= ~(1u << p);
    }
    const v4f *ptr = reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(sc.nodes) + off);
    const v4f q0 = ptr[0], q1 = ptr[1], q2 = ptr[2], q3 = ptr[3], q4 = ptr[4];   // (a leaf record: its first four rows are what a test reads)
    bool stop = false;
    if (!tri) {
        if (COUNT) { cnt.nodes += 4; if (mbcnt64(__ballot(1)) == 0) cnt.wnodes++; }   // 8 boxes = 4 BVH2-equivalent fetches
        // what is left of the group in hand goes to the stack (one entry = two words); the new node's groups take its place
        if ((tv.g_bits & 0xFFu) != 0u) {
            stk.push(tv.sp, (int)tv.g_base); stk.push(tv.sp + STK::unit, (int)tv.g_bits);
            tv.sp += 2 * STK::unit;
        }
        const uint32_t w7 = __float_as_uint(q1.w), w6 = __float_as_uint(q1.z);
        const float ax = q0.w * tv.idx, ay = __uint_as_float(w7 << 16) * tv.idy, az = __uint_as_float(w7 & 0xFFFF0000u) * tv.idz;
        const float bx = q0.x * tv.idx - tv.oix, by = q0.y * tv.idy - tv.oiy, bz = q0.z * tv.idz - tv.oiz;
        const bool negx = (__float_as_uint(tv.idx) >> 31) != 0u, negy = (__float_as_uint(tv.idy) >> 31) != 0u, negz = (__float_as_uint(tv.idz) >> 31) != 0u;
        // rows: q2 = (lo.x[0..3], lo.x[4..7], lo.y[0..3], lo.y[4..7])  q3 = (lo.z.., lo.z.., hi.x.., hi.x..)  q4 = (hi.y.., hi.y.., hi.z.., hi.z..)
        const uint32_t lox0 = __float_as_uint(q2.x), lox1 = __float_as_uint(q2.y), loy0 = __float_as_uint(q2.z), loy1 = __float_as_uint(q2.w);
        const uint32_t loz0 = __float_as_uint(q3.x), loz1 = __float_as_uint(q3.y), hix0 = __float_as_uint(q3.z), hix1 = __float_as_uint(q3.w);
        const uint32_t hiy0 = __float_as_uint(q4.x), hiy1 = __float_as_uint(q4.y), hiz0 = __float_as_uint(q4.z), hiz1 = __float_as_uint(q4.w);
        const uint32_t nx[2] = { negx ? hix0 : lox0, negx ? hix1 : lox1 }, fx[2] = { negx ? lox0 : hix0, negx ? lox1 : hix1 };
        const uint32_t ny[2] = { negy ? hiy0 : loy0, negy ? hiy1 : loy1 }, fy[2] = { negy ? loy0 : hiy0, negy ? loy1 : hiy1 };
        const uint32_t nz[2] = { negz ? hiz0 : loz0, negz ? hiz1 : loz1 }, fz[2] = { negz ? loz0 : hiz0, negz ? loz1 : hiz1 };
        uint32_t miss = 0u;   // after the loop: bit s = slot s was NOT entered (children 7 .. 0 are shifted in from the right)
#pragma unroll
        for (int k = 7; k >= 0; --k) {
            const int h = k >> 2, b = 8 * (k & 3);
            const float tnx = (float)((nx[h] >> b) & 255u) * ax + bx, tfx = (float)((fx[h] >> b) & 255u) * ax + bx;
            const float tny = (float)((ny[h] >> b) & 255u) * ay + by, tfy = (float)((fy[h] >> b) & 255u) * ay + by;
            const float tnz = (float)((nz[h] >> b) & 255u) * az + bz, tfz = (float)((fz[h] >> b) & 255u) * az + bz;
            const float tmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
            const float tmax = fminf(fminf(tfx, tfy), min_raw(tfz, tv.best));
            // entered <=> tmin <= tmax <=> the sign bit of tmax - tmin is clear (equal: +0; an empty slot decodes to an inverted box)
            const float d = tmax - tmin;
            miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d), 31u);   // (miss << 1) | sign(d)
        }
        const uint32_t hit8 = ~miss & 0xFFu, imask = w6 & 0xFFu, lmask = (w6 >> 8) & 0xFFu;
        // inner hits -> priority order: bit p = slot p ^ octinv, i.e. the bit positions XOR-ed with octinv -- three conditional swaps
        uint32_t m = hit8 & imask;
        {
            const uint32_t s1 = ((m & 0xAAu) >> 1) | ((m & 0x55u) << 1); m = (octinv & 1u) ? s1 : m;
            const uint32_t s2 = ((m & 0xCCu) >> 2) | ((m & 0x33u) << 2); m = (octinv & 2u) ? s2 : m;
            const uint32_t s4 = ((m & 0xF0u) >> 4) | ((m & 0x0Fu) << 4); m = (octinv & 4u) ? s4 : m;
        }
        tv.g_base = __float_as_uint(q1.x); tv.g_bits = m | (imask << 8);
        tv.t_base = __float_as_uint(q1.y); tv.t_bits = (hit8 & lmask) | (lmask << 8);
    } else {
        // one leaf record: the strict build tests every triangle by itself; the shipped build decides a QUAD (flag bit 29 of its index word) in one test
        uint32_t flags;
        if (kRecPlanes) {
            const uint32_t iw = (uint32_t)__float_as_int(kRecMetaFirst ? q3.x : 0.0f);
            flags = iw >> 28;
            const bool quad = (flags & 2u) != 0u;
            stop = trav_tri_planes_rec<COUNT>(q0, q1, q2, (uint32_t)__float_as_int(q3.y), (int)(iw & 0x0FFFFFFFu), tv, any, cnt, rec, sc.tris, sc.center, quad, sc.n_rec);
            if (quad) flags = 0u;          // (both triangles decided: nothing more in this leaf)
        } else {
            flags = (uint32_t)__float_as_int(q2.w);
            stop = trav_tri_math<COUNT, true>(q0, q1, q2, tv, any, cnt, rec, sc.tris);
        }
        const bool second = (tv.t_bits & 0x10000u) != 0u;
        const bool more = !second && (flags & 1u) != 0u;      // the leaf's second record is a triangle to test too
        // done with this leaf: clear its slot bit (the lowest set one) and the second-record flag; else: come back for the second record
        tv.t_bits = more ? (tv.t_bits | 0x10000u) : ((tv.t_bits & (tv.t_bits - 1u)) & ~0x10000u);
    }
    // nothing in hand any more: the next group off the stack (the sentinel when nothing is pending)
    if ((tv.t_bits & 0xFFu) == 0u && (tv.g_bits & 0xFFu) == 0u) {
        tv.sp -= 2 * STK::unit;
        const int pb = stk.pop(tv.sp);
        tv.g_bits = (uint32_t)stk.pop(tv.sp + STK::unit);
        tv.g_base = (uint32_t)pb;
        if (pb == kNodeDone) { tv.node = kNodeDone; tv.g_bits = 0u; }
    }
    if (stop) tv.node = kNodeDone;
}
