// experiments/trg_flat.inc.h -- EXPERIMENT, not in the shipped library (TRG_EXPERIMENTS=1 only; TRG_FLAT_PRIMS=1 switches it on per scene load): tiny
// scenes traversed WITHOUT a tree -- every ray tests every primitive of a flat list.  In tolerance; C2 +7 % time, C3 -5 % -- profiles/r05/c2_flat_list_experiment.md.
// Included by toyraygun_amd/csrc/trg_device.h inside namespace trgdev.
#pragma once

// TINY scenes (round 5): no tree walk at all.  Every lane of the wavefront tests EVERY primitive of the flat list, in record order: the loop
// counter and the primitive's 64 bytes are wave-uniform, so the planes arrive by scalar loads from the constant cache and are read as SGPR
// operands -- no stack, no LDS traffic, no divergence: 64 of 64 lanes do the same ~30 instructions per primitive.  For the Cornell box (18
// quads) that is ~540 lane-instructions per ray at full occupancy where the tree needs ~260 per ray at 38 % (the lanes of a wavefront of
// incoherent rays leave the node loop one by one): measured in DESIGN.md section 4.  Same test, same acceptance and tie rule as trav_quad_planes /
// trav_tri_planes; an any-hit query leaves the loop once every lane has its answer.
typedef const __attribute__((address_space(4))) v4f flat_v4f_t;
// inside the unit square <=> max(|s - 1/2|, |t - 1/2|) <= 1/2 (two subtractions, one max with |.| operand modifiers, one compare)
TRG_DEV bool in_unit_square(float s0, float t0) { return fmaxf(fabsf(s0 - 0.5f), fabsf(t0 - 0.5f)) <= 0.5f; }
template <bool ANY, bool COUNT>
TRG_DEV bool traverse_flat(const SceneView &sc, V3 o, V3 d, float tmax_ray, uint32_t rmask, Hit &hit, Counters &cnt) {
    const V3 oc = o - sc.center;
    float best = tmax_ray;
    uint32_t kbest = 0xFFFFFFFFu;       // the primitive of the hit held so far
    // the renderers' rays carry mask 3 or 1: a primitive whose material has bit 0 is then visible to every lane and needs no test per lane
    const bool all_bit0 = __ballot((rmask & 1u) == 0u) == 0ull;   // wave-uniform
    flat_v4f_t *fp = (flat_v4f_t *)(uintptr_t)sc.flat;
    // The loop keeps the distance and the NUMBER of the nearest primitive only; which half of a quad, the weights and the original index are
    // worked out once, afterwards.  The list is in ascending order of original index and is walked from its END with "t <= best": a later
    // (lower-index) primitive at the same distance replaces the held one -- the contract's tie rule -- and a first hit AT the ray's limit counts.
    for (uint32_t k = sc.n_flat; k-- > 0u;) {
        const v4f a = fp[k * 4u], b = fp[k * 4u + 1u], c = fp[k * 4u + 2u], mw = fp[k * 4u + 3u];
        const uint32_t mask = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(mw.z));
        const bool quad = __builtin_amdgcn_readfirstlane(__float_as_int(mw.w)) != 0;
        if (COUNT) { if ((mask & rmask & 3u) != 0u) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
        const float den = a.x * d.x + a.y * d.y + a.z * d.z;
        const float q = a.w - (a.x * oc.x + a.y * oc.y + a.z * oc.z);
        const float t = q * rcp_fast(den);
        const float Px = oc.x + t * d.x, Py = oc.y + t * d.y, Pz = oc.z + t * d.z;
        const float s0 = b.x * Px + b.y * Py + b.z * Pz + b.w;
        const float t0 = c.x * Px + c.y * Py + c.z * Pz + c.w;
        bool ok;
        if (quad) ok = in_unit_square(s0, t0);                  // (scalar branch)
        else ok = (fminf(s0, t0) >= 0.0f) & ((s0 + t0) <= 1.0f);
        ok = ok & (t >= 0.0f) & (t <= best);
        if (!(all_bit0 && (mask & 1u) != 0u)) ok = ok & ((mask & rmask & 3u) != 0u);   // (scalar branch: in a render only the emissive quad takes it)
        best = ok ? t : best;
        kbest = ok ? k : kbest;
        if (ANY && __ballot(kbest == 0xFFFFFFFFu) == 0ull) break;   // every lane of the wavefront has its answer
    }
    const bool found = kbest != 0xFFFFFFFFu;
    hit.t = found ? best : -1.0f; hit.prim = -1; hit.u = 0.0f; hit.v = 0.0f;
    if (!ANY && found) {
        // the winner's own record (per-lane address: vector loads, cached) -> which triangle, its weights and its original index
        const v4f *g = sc.flat + (size_t)kbest * 4u;
        const v4f b = g[1], c = g[2], mw = g[3];
        const float Px = oc.x + best * d.x, Py = oc.y + best * d.y, Pz = oc.z + best * d.z;
        const float s0 = b.x * Px + b.y * Py + b.z * Pz + b.w;
        const float t0 = c.x * Px + c.y * Py + c.z * Pz + c.w;
        const bool quad = __float_as_int(mw.w) != 0;
        const bool second = quad && (s0 < t0);
        hit.u = second ? s0 : (quad ? s0 - t0 : s0);
        hit.v = second ? t0 - s0 : t0;
        hit.prim = __float_as_int(second ? mw.y : mw.x);
    }
    return found;
}

