// q4node.h -- the 64-byte quantised 4-wide BVH node the HBM kernels traverse, and its (host + device) encoder.
//
// Why: the HBM megakernel is bound by the texture-addresser (TA) data path, not by HBM: every 16-byte
// global_load_dwordx4 occupies the TA for 16 cycles per wavefront whatever the lanes address, and a float 4-wide
// node costs seven of them per visit (measured: one more 16-byte load per node = +2.85 ms on the 32.7 ms C4 frame).
// Child boxes quantised to 8 bits against the node's own box bring a node down to four loads and 64 bytes (half
// the node footprint in L2 / Infinity Cache as well).
//
//   dword  0..3   ox, oy, oz, sx        node origin (min corner of the union of the child boxes), x scale
//   dword  4..7   qlo.x qhi.x qlo.y qhi.y   one byte per child (child k in byte k)
//   dword  8..11  qlo.z qhi.z sy sz
//   dword 12..15  child[4]               >= 0: node index, < 0: leaf code ~((first << 3) | (count - 1)), 0x80000000: unused
//
// plane = origin + q * scale, scale = a power of two >= extent / 255.  lo planes are rounded DOWN and hi planes UP
// (computed in double), so the decoded box contains the float box it came from, which already carries the builder's
// 2e-5 x scene-extent padding against the rounding of the slab arithmetic.  Unused slots get the inverted box
// (255, 0) and are skipped by the child check as before.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TRG_Q4_HD __host__ __device__
#else
#define TRG_Q4_HD
#endif

namespace trg {

constexpr uint32_t kQ4NodeBytes = 64;
constexpr int32_t kQ4Empty = (int32_t)0x80000000;

// in: the float 4-wide node (32 floats: lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] child[4] pad[4]); out: 16 dwords
TRG_Q4_HD inline void quantize_node4(const float *in, uint32_t *out) {
    int32_t child[4];
    for (int k = 0; k < 4; ++k) {
        union { float f; int32_t i; } u;
        u.f = in[24 + k];
        child[k] = u.i;
    }
    float origin[3], scale[3];
    uint32_t qlo[3] = { 0, 0, 0 }, qhi[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; ++a) {
        const float *lo = in + a * 8, *hi = in + a * 8 + 4;
        float mn = INFINITY, mx = -INFINITY;
        for (int k = 0; k < 4; ++k)
            if (child[k] != kQ4Empty) { mn = fminf(mn, lo[k]); mx = fmaxf(mx, hi[k]); }
        if (!(mn <= mx)) { mn = 0.0f; mx = 0.0f; }  // no valid child (never produced by the builders)
        const double ext = (double)mx - (double)mn;
        int e = -100;
        if (ext > 0.0) {
            (void)frexp(ext / 255.0, &e);  // ext / 255 = m * 2^e, m in [0.5, 1)  ->  2^e >= ext / 255
            if (e < -100) e = -100;
        }
        for (;;) {
            const double s = ldexp(1.0, e);
            bool ok = true;
            uint32_t pl = 0, ph = 0;
            for (int k = 0; k < 4; ++k) {
                uint32_t l = 255u, h = 0u;  // unused slot: inverted
                if (child[k] != kQ4Empty) {
                    const double dl = floor(((double)lo[k] - (double)mn) / s), dh = ceil(((double)hi[k] - (double)mn) / s);
                    if (dl < 0.0 || dh > 255.0 || !(dl <= 255.0) || !(dh >= 0.0)) { ok = false; break; }
                    l = (uint32_t)dl; h = (uint32_t)dh;
                }
                pl |= l << (8 * k); ph |= h << (8 * k);
            }
            if (ok) { qlo[a] = pl; qhi[a] = ph; origin[a] = mn; scale[a] = (float)s; break; }
            if (++e > 127) {  // non-finite input: give up on precision, never hang (every valid child spans the whole range)
                uint32_t pl = 0, ph = 0;
                for (int k = 0; k < 4; ++k) { const bool v = child[k] != kQ4Empty; pl |= (v ? 0u : 255u) << (8 * k); ph |= (v ? 255u : 0u) << (8 * k); }
                qlo[a] = pl; qhi[a] = ph; origin[a] = mn; scale[a] = 1.0f;
                break;
            }
            // (the float extent was rounded down across a power of two: one coarser step always fits)
        }
    }
    union { float f; uint32_t u; } c;
    c.f = origin[0]; out[0] = c.u; c.f = origin[1]; out[1] = c.u; c.f = origin[2]; out[2] = c.u; c.f = scale[0]; out[3] = c.u;
    out[4] = qlo[0]; out[5] = qhi[0]; out[6] = qlo[1]; out[7] = qhi[1];
    out[8] = qlo[2]; out[9] = qhi[2]; c.f = scale[1]; out[10] = c.u; c.f = scale[2]; out[11] = c.u;
    for (int k = 0; k < 4; ++k) out[12 + k] = (uint32_t)child[k];
}

}  // namespace trg
