// trg_device.h -- device-side building blocks of the gfx950 path-tracing megakernel.
//
// Compiled twice (see build.py): TRG_STRICT=1 with -ffp-contract=off reproduces the CPU oracle's IEEE
// arithmetic operation for operation (parity/debug build); TRG_STRICT=0 is the shipped build (FMA
// contraction, v_rcp/v_rsq in the traversal).  Reference citations are file:line under the reference
// tree; the arithmetic order of every shading function follows the shader source text.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trg.h"
#include "trg_kernels.h"

#ifndef TRG_STRICT
#define TRG_STRICT 0
#endif

#define TRG_DEV __device__ __forceinline__

namespace trgdev {

typedef float v4f __attribute__((ext_vector_type(4)));

struct V3 { float x, y, z; };
TRG_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
TRG_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
TRG_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
TRG_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
TRG_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
TRG_DEV V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
TRG_DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
TRG_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TRG_DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// normalize(v) = v * (1 / sqrt(dot(v,v))): the oracle's definition (oracle/trg_oracle.c normalize3).
TRG_DEV V3 normalize(V3 a) {
#if TRG_STRICT
    const float inv = 1.0f / sqrtf(dot(a, a));
#else
    const float inv = __builtin_amdgcn_rsqf(dot(a, a));
#endif
    return a * inv;
}
TRG_DEV float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
TRG_DEV float rcp_fast(float x) {
#if TRG_STRICT
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
// sqrt / divide of the shading code: IEEE in the strict build (the oracle's arithmetic), the 1-ulp hardware
// instructions in the shipped build (a correctly rounded sqrt or divide is ~10 instructions each on gfx950)
TRG_DEV float sqrt_fast(float x) {
#if TRG_STRICT
    return sqrtf(x);
#else
    return __builtin_amdgcn_sqrtf(x);
#endif
}
TRG_DEV float div_fast(float a, float b) {
#if TRG_STRICT
    return a / b;
#else
    return a * __builtin_amdgcn_rcpf(b);
#endif
}

// ---------------------------------------------------------------------------------------------
// Halton (a5): runtime/shaders/common.h:51-75, prime table extended per SURVEY F5.
//   reference recurrence:  f = f * invB;  r = r + f * (i % b);  i = i / b;   while (i > 0)
// Device form: the base is a compile-time constant, so f_k = fl(f_{k-1} * fl(1/b)) is a literal; the loop
// runs the fixed digit count of a 32-bit index: once the quotient reaches 0 the remaining terms add
// f_k * 0 = +0, which leaves r unchanged, so the value is bit-identical to the data-dependent loop.
// Never contracted, in either build.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr uint32_t halton_prime(uint32_t d) {
    constexpr uint32_t p[64] = { 2,   3,   5,   7,   11,  13,  17,  19,  23,  29,  31,  37,  41,  43,  47,  53,
                                 59,  61,  67,  71,  73,  79,  83,  89,  97,  101, 103, 107, 109, 113, 127, 131,
                                 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223,
                                 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311 };
    return p[d & 63];
}
__host__ __device__ constexpr int halton_digits(uint32_t b) {
    int n = 0;
    uint64_t v = 1;
    while (v <= 0xFFFFFFFFull) { v *= b; ++n; }
    return n;  // smallest n with b^n > 2^32-1
}
template <uint32_t B>
struct HaltonTab {
    float f[33];
    constexpr HaltonTab() : f{} {
        const float invB = 1.0f / (float)B;
        float v = 1.0f;
        for (int k = 0; k < 33; ++k) { v = v * invB; f[k] = v; }
    }
};
template <uint32_t B>
struct HaltonRcpPow {
    float f[24];  // f[k] = fl(1 / b^k) for the powers b^k <= 2^22 (b^k is exact in fp32), 0 beyond
    constexpr HaltonRcpPow() : f{} {
        uint64_t v = 1;
        for (int k = 0; k < 24; ++k) {
            f[k] = v <= (1ull << 22) ? 1.0f / (float)v : 0.0f;
            v = v * B > (1ull << 40) ? (1ull << 40) : v * B;
        }
    }
};
template <uint32_t B>
struct HaltonWeight {
    float f[33];  // f[k] = b^-k, correctly rounded (weight of a digit group that starts at digit k)
    constexpr HaltonWeight() : f{} {
        double v = 1.0;
        for (int k = 0; k < 33; ++k) { f[k] = (float)v; v /= (double)B; }
    }
};
// Device evaluation of one dimension, bit-identical to the reference loop (validated exhaustively on the
// CPU by tests/helpers/halton_identities.c):
//  * base 2: closed form.  The running sum keeps the exponent of its first (largest) term, so every later
//    term is either a multiple of the ulp (added exactly), exactly half an ulp (one round-to-even tie) or
//    below half an ulp (dropped): bit-reverse, keep 24 bits from the leading one, resolve the single tie.
//  * odd primes: ONE integer division splits i = hi * b^k1 + lo with lo, hi < 2^22; below 2^22 the digits
//    come out of pure fp32 arithmetic -- q = floor((n + 0.5) * fl(1/b)) is exactly n div b there, and
//    d = fma(q, -b, n) is exact -- six full-rate VALU instructions per digit, every constant a literal or
//    an inline operand (no registers), instead of two quarter-rate integer multiplies.  (A fused
//    fma(n, 1/b, 0.5/b) would save one instruction but needs two non-inline constants, which gfx950's
//    one-slot constant bus only allows with a VGPR per base: that cost 60 registers.)  The products
//    f_k * d and the sums r + t are never contracted.
__host__ __device__ constexpr int halton_lo_digits(uint32_t b) {
    int k = 0;
    uint64_t v = 1;
    while (v * b <= (1ull << 22)) { v *= b; ++k; }
    return k;  // largest k with b^k <= 2^22
}
__host__ __device__ constexpr uint32_t halton_pow(uint32_t b, int k) {
    uint32_t v = 1;
    for (int j = 0; j < k; ++j) v *= b;
    return v;
}
template <uint32_t D>
TRG_DEV float halton_c(uint32_t i) {
#pragma clang fp contract(off)
#ifdef TRG_EXP_NOHALTON  // timing-only ablation: a cheap hash instead of the radical inverse
    return (float)((i * 2654435761u + D * 40503u) >> 8) * 0x1p-24f;
#endif
    constexpr uint32_t B = halton_prime(D);
    if constexpr (B == 2) {
        const uint32_t rev = __brev(i);
        const uint32_t lz = rev ? (uint32_t)__clz((int)rev) : 32u;
        const uint32_t drop = lz < 8u ? 8u - lz : 0u;
        const uint32_t kept = (rev >> drop) << drop;
        float r = (float)kept * 0x1p-32f;
        if (drop) {
            const uint32_t tie = (rev >> (drop - 1u)) & 1u, lsb = (rev >> drop) & 1u;
            if (tie & lsb) r = r + __uint_as_float((127u - lz - 24u) << 23);
        }
        return r;
    } else {
        constexpr int N = halton_digits(B);
        constexpr int K1 = halton_lo_digits(B);
        constexpr uint32_t B1 = halton_pow(B, K1);
        constexpr HaltonTab<B> tab{};
        constexpr HaltonRcpPow<B> rp{};
        constexpr float bf = (float)B;
        const uint32_t hi = i / B1;
        const uint32_t lo = i - hi * B1;
        float r = 0.0f;
        // Every quotient comes straight from n -- q_k = floor((n + 0.5) * fl(1 / b^k)) = n div b^k exactly for n < 2^22
        // (same error bound as for k = 1; checked exhaustively in tests/helpers/halton_identities.c) -- instead of from
        // the previous quotient, so the digits d_k = q_k - b * q_{k+1} are independent of one another: no serial
        // add -> mul -> floor chain per digit, and one instruction less.  The sum still runs in digit order.
        {
            const float n = (float)lo, nh = n + 0.5f;
            float q0 = n;
#pragma unroll
            for (int k = 0; k < K1; ++k) {
                const float q1 = (k + 1 < K1) ? __builtin_floorf(nh * rp.f[k + 1]) : 0.0f;  // lo < b^K1
                const float dig = __builtin_fmaf(q1, -bf, q0);
                const float t = tab.f[k] * dig;
                r = r + t;
                q0 = q1;
            }
        }
        {
            const float n = (float)hi, nh = n + 0.5f;
            float q0 = n;
#pragma unroll
            for (int k = 0; k < N - K1; ++k) {
                const float q1 = (k + 1 < N - K1) ? __builtin_floorf(nh * rp.f[k + 1]) : 0.0f;  // hi < b^(N-K1)
                const float dig = __builtin_fmaf(q1, -bf, q0);
                const float t = tab.f[K1 + k] * dig;
                r = r + t;
                q0 = q1;
            }
        }
        return r;
    }
}
// Table form of dimensions 1..5 (bases 3, 5, 7, 11, 13) for the shipped megakernel on an LDS-resident scene: the same
// hi / lo split and the same direct quotients, but in radix R = b^K -- one LDS lookup of the radical inverse of a whole
// K-digit group (trg_kernels.h kHtab) instead of K digits.  Not bit-identical to the digit loop (the groups are summed
// with correctly rounded weights), within an ulp or two of it; the strict build never uses it.
// CHECK (test entry trg_halton_table only): a group remainder outside [0, radix) -- an out-of-range table index -- poisons the
// result with NaN; the arithmetic is otherwise the megakernel's, instruction for instruction.
template <uint32_t D, bool CHECK = false>
TRG_DEV float halton_t(uint32_t i, const float *htab) {
    constexpr uint32_t B = halton_prime(D);
    constexpr trg::HtabSpec spec = trg::kHtab[D - 1];
    static_assert(D >= 1 && D <= 5 && spec.base == B, "table dimensions are 1..5");
    constexpr int K = (int)spec.digits;
    constexpr int N = halton_digits(B);
    constexpr int K1 = halton_lo_digits(B);
    constexpr uint32_t B1 = halton_pow(B, K1);
    constexpr HaltonRcpPow<B> rp{};
    constexpr HaltonWeight<B> wt{};
    constexpr float Rf = (float)spec.radix;
    const float *T = htab + spec.offset;
    const uint32_t hi = i / B1;
    const uint32_t lo = i - hi * B1;
    float r = 0.0f;
    bool bad = false;
    {
        constexpr int G = (K1 + K - 1) / K;
        const float n = (float)lo, nh = n + 0.5f;
        float q0 = n;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float q1 = (g + 1 < G) ? __builtin_floorf(nh * rp.f[(g + 1) * K]) : 0.0f;
            const float rem = __builtin_fmaf(q1, -Rf, q0);
            if (CHECK && !(rem >= 0.0f && rem < Rf)) bad = true;
            else r = __builtin_fmaf(T[(int)rem], wt.f[g * K], r);
            q0 = q1;
        }
    }
    {
        constexpr int G = (N - K1 + K - 1) / K;
        const float n = (float)hi, nh = n + 0.5f;
        float q0 = n;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float q1 = (g + 1 < G) ? __builtin_floorf(nh * rp.f[(g + 1) * K]) : 0.0f;
            const float rem = __builtin_fmaf(q1, -Rf, q0);
            if (CHECK && !(rem >= 0.0f && rem < Rf)) bad = true;
            else r = __builtin_fmaf(T[(int)rem], wt.f[K1 + g * K], r);
            q0 = q1;
        }
    }
    return (CHECK && bad) ? __builtin_nanf("") : r;
}
// any dimension the way the shipped megakernel evaluates it on an LDS-resident scene: group tables for 1..5, digits otherwise
TRG_DEV float halton_any(uint32_t i, uint32_t d);
TRG_DEV float halton_any_tab(uint32_t i, uint32_t d, const float *htab) {
    switch (d & 63u) {
    case 1: return halton_t<1, true>(i, htab);
    case 2: return halton_t<2, true>(i, htab);
    case 3: return halton_t<3, true>(i, htab);
    case 4: return halton_t<4, true>(i, htab);
    case 5: return halton_t<5, true>(i, htab);
    default: return halton_any(i, d);
    }
}
template <uint32_t D, bool TAB>
TRG_DEV float halton_d(uint32_t i, const float *htab) {
    if constexpr (TAB && D >= 1 && D <= 5) return halton_t<D>(i, htab);
    else return halton_c<D>(i);
}
// the two raygen dimensions (Raytracing.metal:67-68)
template <bool TAB = false>
TRG_DEV void halton_pixel(uint32_t i, float &r0, float &r1, const float *htab = nullptr) { r0 = halton_c<0>(i); r1 = halton_d<1, TAB>(i, htab); }
// the four dimensions 2 + 4*bounce + {0,1,2,3} of one shading event (Raytracing.metal:165-166,188-189);
// `bounce` is wave-uniform so this is a scalar branch.
template <int BNC, bool TAB = false>
TRG_DEV void halton4_c(uint32_t i, float r[4], const float *htab = nullptr) {
    r[0] = halton_d<2 + 4 * BNC + 0, TAB>(i, htab);
    r[1] = halton_d<2 + 4 * BNC + 1, TAB>(i, htab);
    r[2] = halton_d<2 + 4 * BNC + 2, TAB>(i, htab);
    r[3] = halton_d<2 + 4 * BNC + 3, TAB>(i, htab);
}
// only the two light-sample dimensions 2 + 4*bounce + {0,1}: on the last bounce the continuation ray is never
// traced, so its two dimensions (and the hemisphere sample built from them) are not evaluated at all
template <int BNC, bool TAB = false>
TRG_DEV void halton2_c(uint32_t i, float r[4], const float *htab = nullptr) {
    r[0] = halton_d<2 + 4 * BNC + 0, TAB>(i, htab);
    r[1] = halton_d<2 + 4 * BNC + 1, TAB>(i, htab);
}
template <bool TAB = false>
TRG_DEV void halton2(uint32_t i, uint32_t bounce, float r[4], const float *htab = nullptr) {
    switch (bounce) {
    case 0: halton2_c<0, TAB>(i, r, htab); break;
    case 1: halton2_c<1>(i, r); break;
    case 2: halton2_c<2>(i, r); break;
    case 3: halton2_c<3>(i, r); break;
    case 4: halton2_c<4>(i, r); break;
    case 5: halton2_c<5>(i, r); break;
    case 6: halton2_c<6>(i, r); break;
    case 7: halton2_c<7>(i, r); break;
    case 8: halton2_c<8>(i, r); break;
    case 9: halton2_c<9>(i, r); break;
    case 10: halton2_c<10>(i, r); break;
    case 11: halton2_c<11>(i, r); break;
    case 12: halton2_c<12>(i, r); break;
    case 13: halton2_c<13>(i, r); break;
    default: halton2_c<14>(i, r); break;
    }
}
template <bool TAB = false>
TRG_DEV void halton4(uint32_t i, uint32_t bounce, float r[4], const float *htab = nullptr) {
    switch (bounce) {
    case 0: halton4_c<0, TAB>(i, r, htab); break;
    case 1: halton4_c<1>(i, r); break;
    case 2: halton4_c<2>(i, r); break;
    case 3: halton4_c<3>(i, r); break;
    case 4: halton4_c<4>(i, r); break;
    case 5: halton4_c<5>(i, r); break;
    case 6: halton4_c<6>(i, r); break;
    case 7: halton4_c<7>(i, r); break;
    case 8: halton4_c<8>(i, r); break;
    case 9: halton4_c<9>(i, r); break;
    case 10: halton4_c<10>(i, r); break;
    case 11: halton4_c<11>(i, r); break;
    case 12: halton4_c<12>(i, r); break;
    case 13: halton4_c<13>(i, r); break;
    default: halton4_c<14>(i, r); break;
    }
}
// any dimension, per-lane (test entry trg_halton): the same halton_c<D> instantiations the megakernel uses
TRG_DEV float halton_any(uint32_t i, uint32_t d) {
    switch (d & 63u) {
    case 0: return halton_c<0>(i);
    case 1: return halton_c<1>(i);
    case 2: return halton_c<2>(i);
    case 3: return halton_c<3>(i);
    case 4: return halton_c<4>(i);
    case 5: return halton_c<5>(i);
    case 6: return halton_c<6>(i);
    case 7: return halton_c<7>(i);
    case 8: return halton_c<8>(i);
    case 9: return halton_c<9>(i);
    case 10: return halton_c<10>(i);
    case 11: return halton_c<11>(i);
    case 12: return halton_c<12>(i);
    case 13: return halton_c<13>(i);
    case 14: return halton_c<14>(i);
    case 15: return halton_c<15>(i);
    case 16: return halton_c<16>(i);
    case 17: return halton_c<17>(i);
    case 18: return halton_c<18>(i);
    case 19: return halton_c<19>(i);
    case 20: return halton_c<20>(i);
    case 21: return halton_c<21>(i);
    case 22: return halton_c<22>(i);
    case 23: return halton_c<23>(i);
    case 24: return halton_c<24>(i);
    case 25: return halton_c<25>(i);
    case 26: return halton_c<26>(i);
    case 27: return halton_c<27>(i);
    case 28: return halton_c<28>(i);
    case 29: return halton_c<29>(i);
    case 30: return halton_c<30>(i);
    case 31: return halton_c<31>(i);
    case 32: return halton_c<32>(i);
    case 33: return halton_c<33>(i);
    case 34: return halton_c<34>(i);
    case 35: return halton_c<35>(i);
    case 36: return halton_c<36>(i);
    case 37: return halton_c<37>(i);
    case 38: return halton_c<38>(i);
    case 39: return halton_c<39>(i);
    case 40: return halton_c<40>(i);
    case 41: return halton_c<41>(i);
    case 42: return halton_c<42>(i);
    case 43: return halton_c<43>(i);
    case 44: return halton_c<44>(i);
    case 45: return halton_c<45>(i);
    case 46: return halton_c<46>(i);
    case 47: return halton_c<47>(i);
    case 48: return halton_c<48>(i);
    case 49: return halton_c<49>(i);
    case 50: return halton_c<50>(i);
    case 51: return halton_c<51>(i);
    case 52: return halton_c<52>(i);
    case 53: return halton_c<53>(i);
    case 54: return halton_c<54>(i);
    case 55: return halton_c<55>(i);
    case 56: return halton_c<56>(i);
    case 57: return halton_c<57>(i);
    case 58: return halton_c<58>(i);
    case 59: return halton_c<59>(i);
    case 60: return halton_c<60>(i);
    case 61: return halton_c<61>(i);
    case 62: return halton_c<62>(i);
    case 63: return halton_c<63>(i);
    }
    return 0.0f;
}

// ---------------------------------------------------------------------------------------------
// sin/cos of phi in [0, 2*pi] (common.h:82-85).  The reference calls Metal's cos()/sin(), whose
// rounding is implementation-defined; the project's device definition is this fixed polynomial,
// restated operation for operation by the oracle (orc_sincos_portable).  Never contracted.
// ---------------------------------------------------------------------------------------------
TRG_DEV void trg_sincos(float phi, float &s_out, float &c_out) {
#pragma clang fp contract(off)
    const float two_over_pi = 0x1.45f306p-1f;
    const float pio2_hi = 0x1.92p+0f;
    const float pio2_mid = 0x1.fb5444p-12f;
    const float pio2_lo = 0x1.68c2p-39f;
    const float kf = rintf(phi * two_over_pi);
    const int k = (int)kf;
    float r = phi - kf * pio2_hi;
    r = r - kf * pio2_mid;
    r = r - kf * pio2_lo;
    const float z = r * r;
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    const float sn = (ps * z) * r + r;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    const float cs = ((pc * z) * z - 0.5f * z) + 1.0f;
    const int q = k & 3;
    const float s = (q == 0) ? sn : (q == 1) ? cs : (q == 2) ? -sn : -cs;
    const float c = (q == 0) ? cs : (q == 1) ? -sn : (q == 2) ? -cs : sn;
    s_out = s;
    c_out = c;
}

// common.h:80-91
TRG_DEV V3 sample_cosine_hemisphere(float ux, float uy) {
    const float PI_F = 3.1415926535898f;
    float sin_phi, cos_phi;
#if TRG_STRICT
    const float phi = 2.0f * PI_F * ux;
    trg_sincos(phi, sin_phi, cos_phi);
#else
    // shipped build: v_sin_f32 / v_cos_f32 take their argument in revolutions, which is exactly ux -- two
    // transcendental instructions instead of the range reduction and two polynomials
    (void)PI_F;
    sin_phi = __builtin_amdgcn_sinf(ux);
    cos_phi = __builtin_amdgcn_cosf(ux);
#endif
    const float cos_theta = sqrt_fast(uy);
    const float sin_theta = sqrt_fast(1.0f - cos_theta * cos_theta);
    return mk(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
}
// common.h:95-110
TRG_DEV V3 align_hemisphere(V3 s, V3 n) {
    const V3 up = n;
    const V3 right = normalize(cross(n, mk(0.0072f, 1.0f, 0.0034f)));
    const V3 forward = cross(right, up);
    return s.x * right + s.y * up + s.z * forward;
}
struct LightSample { V3 dir; V3 color; float dist; };
// common.h:119-159
TRG_DEV LightSample sample_area_light(const trg_uniforms &u, float r0, float r1, V3 p, V3 n) {
    LightSample ls;
    const float ux = r0 * 2.0f - 1.0f, uy = r1 * 2.0f - 1.0f;
    const V3 lp = mk(u.light_pos[0], u.light_pos[1], u.light_pos[2]);
    const V3 lr = mk(u.light_right[0], u.light_right[1], u.light_right[2]);
    const V3 lu = mk(u.light_up[0], u.light_up[1], u.light_up[2]);
    const V3 lf = mk(u.light_forward[0], u.light_forward[1], u.light_forward[2]);
    const V3 sp = lp + lr * ux + lu * uy;
    V3 d = sp - p;
    const float dist = sqrt_fast(dot(d, d));
    const float inv = div_fast(1.0f, fmaxf(dist, 1e-3f));
    d = d * inv;
    V3 c = mk(u.light_color[0], u.light_color[1], u.light_color[2]);
    c = c * (inv * inv);
    c = c * saturate(dot(neg(d), lf));
    c = c * saturate(dot(n, d));
    ls.dir = d; ls.color = c; ls.dist = dist;
    return ls;
}

// Raytracing.metal:41-91 (a6)
template <bool TAB = false>
TRG_DEV void raygen(const trg_uniforms &u, uint32_t x, uint32_t y, uint32_t hidx, V3 &org, V3 &dir, const float *htab = nullptr) {
    float r0, r1;
    halton_pixel<TAB>(hidx, r0, r1, htab);
    const float px = (float)x + r0, py = (float)y + r1;
    // Opaque copies: 1/width, 1/height and the four 0*m products below are wave-uniform and loop-invariant, so LICM hoists them
    // out of the megakernel's frame loop -- as VALU results they then sit in six VGPRs across every traversal and were
    // spilled to scratch at 8 waves/SIMD.  Recomputing them per frame is eight instructions.  Same arithmetic, same bits.
    uint32_t wv = u.width, hv = u.height;
    float zero = 0.0f;
    asm volatile("" : "+v"(wv), "+v"(hv), "+v"(zero));
    float uvx = div_fast(px, (float)wv), uvy = div_fast(py, (float)hv);
    uvx = uvx * 2.0f - 1.0f;
    uvy = uvy * 2.0f - 1.0f;
    const float *m = u.inv_view_proj;
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = uvx * m[j * 4 + 0] + uvy * m[j * 4 + 1] + zero * m[j * 4 + 2] + 1.0f * m[j * 4 + 3];
    const V3 cam = mk(u.cam_pos[0], u.cam_pos[1], u.cam_pos[2]);
    const V3 world = mk(div_fast(w[0], w[3]), div_fast(w[1], w[3]), div_fast(w[2], w[3]));
    org = cam;
    dir = normalize(world - cam);
}

// ---------------------------------------------------------------------------------------------
// Intersector (a7 nearest, a12 any): replaces MPSRayIntersector (MetalRenderer.mm:265-279,427-437,
// 466-475).  Contract = oracle/trg_oracle.c "INTERSECTION CONTRACT".
// ---------------------------------------------------------------------------------------------
// A scene staged in LDS keeps the reference's buffers: 48-byte triangle records in leaf order + normals / colours / material ids
// in ORIGINAL order (Raytracing.metal:104-108).  A scene traversed from HBM has ONE 128-byte record per triangle, in leaf order
// (kRecV4 float4 each, a cache line): rows 0..2 the same geometry record, rows 3..7 the nine normal and nine colour floats; the
// material id is the mask of row 1.  A hit then names its RECORD (Hit::prim = record index), the shading event reads the line the
// triangle test has just pulled through L1 / L2 instead of three more lines of three other arrays (two dependent round trips
// through memory per shading event before), and the original index is only needed to break distance ties (trav_tri_math).
constexpr int kRecV4 = 8;   // float4 per leaf record of an HBM-resident scene
struct SceneView {
    const v4f *nodes;       // 4 per node
    const v4f *tris;        // LDS scene: 3 per triangle; HBM scene: kRecV4 per triangle (geometry + attributes); leaf order
    const float *normals;   // LDS scene: 9 per triangle, ORIGINAL order (reference vertexNormals buffer); HBM scene: nullptr
    const float *colors;    // LDS scene: 9 per triangle, ORIGINAL order (reference vertexColors buffer); HBM scene: nullptr
    const uint32_t *mats;   // LDS scene: 1 per triangle, ORIGINAL order (reference triangleMasks buffer); HBM scene: nullptr
    const float *htab;      // Halton group tables in LDS (trg_kernels.h kHtab), or nullptr
    trg::TexDesc tex;       // albedo textures in global memory (tex.uv == nullptr: none)
    const unsigned short *meta;   // LDS scene: per record (original index << 2) | (material id & 3) (the plane test of the shipped build)
    V3 center;              // shipped build: the point the plane records are relative to (SceneDesc::center); Trav::o is relative to it too
    uint32_t rec_delta;     // HBM scene: byte distance from `nodes` to `tris` (the records follow the nodes in the blob)
    uint32_t n_rec;         // HBM scene: its leaf records; an index from here on names a BOX record (shipped build: SceneDesc::off_boxrec sits right behind them)
    const v4f *flat;        // shipped build, tiny LDS-resident scene: the flat primitive list in global memory (trg_kernels.h SceneDesc::off_flat), wave-uniform
    uint32_t n_flat;        // its entries; 0: walk the tree
};
// original index / material id of leaf record r of an HBM-resident scene: the .w of rows 0 / 1 (Moeller-Trumbore records), or the last two
// words of the record (plane records, whose rows 0..2 are full)
TRG_DEV int fat_prim(const v4f *recs, uint32_t r);
TRG_DEV uint32_t fat_mask(const v4f *recs, uint32_t r);
struct Hit { float t; int prim; float u, v; };  // u, v = Moeller-Trumbore weights of vertex 1 and 2; prim = original index (LDS scene) or leaf record (HBM scene)
struct Counters { uint32_t nodes, tris, wnodes, wtris; };  // per-lane work and wave-level iterations (first active lane counts)

// One ray/triangle test.  Returns true when (u,v) are inside and t is in [0, tmax_ray].
TRG_DEV bool tri_test(const v4f a, const v4f b, const v4f c, V3 o, V3 d, float tmax_ray, float &t, float &u, float &v) {
    const V3 v0 = mk(a.x, a.y, a.z), e1 = mk(b.x, b.y, b.z), e2 = mk(c.x, c.y, c.z);
    const V3 p = cross(d, e2);
    const float det = dot(e1, p);
    const float inv = rcp_fast(det);
    const V3 tv = o - v0;
    u = dot(tv, p) * inv;
    const V3 q = cross(tv, e1);
    v = dot(d, q) * inv;
    t = dot(e2, q) * inv;
    return (u >= 0.0f) && (v >= 0.0f) && ((u + v) <= 1.0f) && (t >= 0.0f) && (t <= tmax_ray);
}

// The same test on a triangle given as THREE PLANES (shipped build, LDS-resident scenes; trg_capi.cpp fill_plane_record): a = (n, d0) the
// unit supporting plane, b = (n1, d1) and c = (n2, d2) the planes whose signed distances are the weights of vertex 1 and 2.
// t = (d0 - n.o) / (n.d), P = o + t d, u = n1.P + d1, v = n2.P + d2: 17 arithmetic instructions against Moeller-Trumbore's 31, the same
// (t, u, v) up to rounding.  A ray parallel to the plane gets t = +-inf or NaN and then u, v = NaN: no hit, like MT's det = 0.
TRG_DEV bool tri_test_planes(const v4f a, const v4f b, const v4f c, V3 o, V3 d, float tmax_ray, float &t, float &u, float &v) {
    const float den = a.x * d.x + a.y * d.y + a.z * d.z;
    const float q = a.x * o.x + (a.y * o.y + (a.z * o.z - a.w));
    t = -q * rcp_fast(den);
    const V3 P = mk(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
    u = b.x * P.x + (b.y * P.y + (b.z * P.z + b.w));
    v = c.x * P.x + (c.y * P.y + (c.z * P.z + c.w));
    return (u >= 0.0f) && (v >= 0.0f) && ((u + v) <= 1.0f) && (t >= 0.0f) && (t <= tmax_ray);
}
// The same planes for a QUAD leaf (bvh_build.h): a / b / c are the parallelogram's -- P = p0 + s X.e1 + t Y.e2 -- and ONE test decides both
// triangles: inside iff 0 <= s, t <= 1; s >= t is triangle X (record `first`) with weights (u, v) = (s - t, t), t > s triangle Y (the next
// record) with (s, t - s).  `second` says which.  quad = false: the plain triangle test (inside iff s + t <= 1).
TRG_DEV bool tri_test_planes_quad(const v4f a, const v4f b, const v4f c, V3 o, V3 d, float tmax_ray, bool quad, float &t, float &u, float &v, bool &second) {
    const float den = a.x * d.x + a.y * d.y + a.z * d.z;
    const float q = a.x * o.x + (a.y * o.y + (a.z * o.z - a.w));
    t = -q * rcp_fast(den);
    const V3 P = mk(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
    const float s0 = b.x * P.x + (b.y * P.y + (b.z * P.z + b.w));
    const float t0 = c.x * P.x + (c.y * P.y + (c.z * P.z + c.w));
    second = quad && (s0 < t0);
    const bool in = (s0 >= 0.0f) && (t0 >= 0.0f) && (quad ? ((s0 <= 1.0f) && (t0 <= 1.0f)) : ((s0 + t0) <= 1.0f));
    u = second ? s0 : (quad ? s0 - t0 : s0);
    v = second ? t0 - s0 : t0;
    return in && (t >= 0.0f) && (t <= tmax_ray);
}
constexpr uint32_t kLeafQuad = 7u;   // bvh_build.h: the count field of a QUAD leaf (two triangles of a parallelogram, X then Y)
constexpr uint32_t kLeafBox = 6u;    // ... of a BOX leaf (LDS-resident scenes): twelve records, six quads that bound a parallelepiped
// the build's triangle test on an LDS-resident scene: planes + the u16 per record (shipped), or the Moeller-Trumbore rows (strict)
constexpr bool kTriPlanes = !TRG_STRICT && TRG_TRI_PLANES;
// ... and on a scene traversed from HBM: the leaf records of SceneDesc::off_fat_planes (rows 0..2 planes, words 30 / 31 index and mask)
constexpr bool kRecPlanes = !TRG_STRICT && TRG_TRI_PLANES_HBM;

constexpr bool kRecMetaFirst = kRecPlanes && TRG_REC_META_FIRST;   // plane records: index / mask = floats 12, 13 (else 30, 31)
constexpr bool kBoxHbm = kRecPlanes && TRG_BOX_LEAVES && TRG_BOX_LEAVES_HBM;   // box leaves in the tree traversed from HBM (shipped build only)
static_assert(!kBoxHbm || kRecMetaFirst, "a box record keeps its first record, mask and face table in row 3, which the unified step loads for TRG_REC_META_FIRST");
// (TRG_WIDE8: the index word of a record carries two flag bits, 28 and 29 -- trav_step_wide8)
TRG_DEV int fat_prim(const v4f *recs, uint32_t r) { return (TRG_WIDE8 ? 0x0FFFFFFF : -1) & __float_as_int(kRecMetaFirst ? recs[(size_t)r * kRecV4 + 3].x : kRecPlanes ? recs[(size_t)r * kRecV4 + 7].z : recs[(size_t)r * kRecV4].w); }
TRG_DEV uint32_t fat_mask(const v4f *recs, uint32_t r) { return (uint32_t)__float_as_int(kRecMetaFirst ? recs[(size_t)r * kRecV4 + 3].y : kRecPlanes ? recs[(size_t)r * kRecV4 + 7].w : recs[(size_t)r * kRecV4 + 1].w); }

// Per-lane traversal stack, laid out [level][thread] so lane i always hits LDS bank i%32 (no conflicts).
// Scenes staged in LDS have shallow trees and keep the whole stack in LDS (klds = all levels).  Scenes in
// HBM keep the first klds levels in LDS and spill deeper levels to a per-thread column of a global scratch
// buffer: the 4-wide tree can need 3 x depth entries in the worst case, typical rays need about a dozen,
// and LDS spent on never-used levels is occupancy lost.
// The stack pointer a traversal carries is opaque: stk.first() is the empty stack, it moves by STK::unit per entry.
// It IS the LDS byte address of the next free entry (unit = BLOCK ints), so push and pop are a bare ds_write_b32 / ds_read_b32
// with no address arithmetic; with overflow one compare against a wave-uniform limit tells the levels that live in the scratch
// (whose index is then recovered from the address: rare).
typedef __attribute__((address_space(3))) int lds_int_t;
constexpr int kNodeDone = (int)0x80000000;  // "traversal finished" marker in the node register (never a valid leaf code)
// Level 0 of every stack holds a permanent SENTINEL (kNodeDone, written once per thread by init()): a traversal starts with its
// stack pointer at level 1, and the pop that would underflow returns the sentinel, which ends the traversal -- so pop never tests
// for an empty stack (two instructions less per pop, and neither the marker nor the stack base has to stay in a VGPR).
template <int BLOCK, bool OVERFLOW = false>
struct LdsStackT {
    uint32_t lds;      // LDS byte address of level 0 of this thread's column (already offset by the thread index)
    int *overflow;     // global scratch for the levels beyond klds (wave-uniform: stays in SGPRs), or nullptr
    uint32_t gcol;     // this thread's column in the scratch: workgroup * BLOCK + thread
    uint32_t gstride;  // elements between consecutive overflow levels
    uint32_t lim;      // LDS byte address of level klds of thread 0's column (wave-uniform).  A column is BLOCK * 4 bytes wide, so for
                       // every thread `sp >= lim` <=> level >= klds, and (sp - lim) / (BLOCK * 4) is the level's index in the scratch
    static constexpr int unit = BLOCK * 4;
    TRG_DEV void set(unsigned char *smem, uint32_t stack_off, int *ovf, uint32_t klds_) {
        const uint32_t col0 = (uint32_t)(uintptr_t)(lds_int_t *)reinterpret_cast<int *>(smem + stack_off);
        lds = col0 + threadIdx.x * 4u;
        overflow = ovf; gcol = blockIdx.x * BLOCK + threadIdx.x; gstride = gridDim.x * BLOCK; lim = col0 + klds_ * (uint32_t)(BLOCK * 4);
        *(lds_int_t *)(uintptr_t)lds = kNodeDone;   // the sentinel at level 0
    }
    // (TRG_WIDE8: the stack of an HBM traversal holds PAIRS of words -- a node group -- so the empty stack stands two levels up and the pop
    //  that would underflow reads the sentinel at level 0 and whatever is at level 1)
    TRG_DEV int first() const { return (int)lds + BLOCK * 4 * ((TRG_WIDE8 && OVERFLOW) ? 2 : 1); }
    TRG_DEV void push(int sp, int v) {
        if (!OVERFLOW || (uint32_t)sp < lim) *(lds_int_t *)(uintptr_t)(uint32_t)sp = v;
        else overflow[(size_t)(((uint32_t)sp - lim) / (uint32_t)(BLOCK * 4) * gstride + gcol)] = v;
    }
    // room(sp, n): the n entries from sp on are all in LDS; put(): a bare LDS store for such an entry (the branch-free pushes of wide_select)
    TRG_DEV bool room(int sp, int n) const { return !OVERFLOW || (uint32_t)sp + (uint32_t)(n * unit) <= lim; }
    TRG_DEV void put(int sp, int v) { *(lds_int_t *)(uintptr_t)(uint32_t)sp = v; }
    TRG_DEV int pop(int sp) {
        if (!OVERFLOW || (uint32_t)sp < lim) return *(lds_int_t *)(uintptr_t)(uint32_t)sp;
        return overflow[(size_t)(((uint32_t)sp - lim) / (uint32_t)(BLOCK * 4) * gstride + gcol)];
    }
};

TRG_DEV uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Per-ray traversal state, resumable: the queue tracer parks a ray in these registers while other lanes refill.
struct Trav {
    V3 o, d;
    float idx, idy, idz, oix, oiy, oiz;  // reciprocal direction and origin * reciprocal direction
    float best;        // the ray's maxDistance until something is hit, then the distance of the accepted hit (any-hit: of the hit that ends it)
    uint32_t rmask;
    Hit hit;           // prim, u, v of the accepted hit; hit.t is filled in by trav_hit() only (it is `best`)
    bool found;
    int node, sp;
#if TRG_WIDE8
    // compressed 8-wide tree (q8node.h): the node GROUP in hand -- the inner children of one node that were hit and are still to be entered:
    // g_base = index of that node's first inner child, g_bits = hit mask in PRIORITY order (bit p <-> slot p ^ octinv; the highest goes first) |
    // the node's inner-slot mask << 8 -- and its leaf group: t_base = first leaf record of the node, t_bits = hit leaf slots | leaf-slot mask << 8 |
    // (bit 16: the SECOND record of the lowest pending leaf is next)
    uint32_t g_base, g_bits, t_base, t_bits;
#endif
    int sx, sz;      // sign-ordered LDS nodes: LDS address of this ray's [X | Y] block and of its Z block in node 0 (added to a node's byte offset)
};

TRG_DEV float clamp_away_from_zero(float v) {
    const uint32_t b = __float_as_uint(v);
    const uint32_t mag = b & 0x7fffffffu, tiny = 0x0da24260u;  // 1e-30f
    return __uint_as_float((b & 0x80000000u) | (mag < tiny ? tiny : mag));
}
// The plane tests want the ray origin RELATIVE to the scene's centre (SceneView::center).  A traversal of an LDS-resident scene (`rel`) keeps
// o - centre in Trav::o -- nothing but the triangle test reads it there, the kernels keep the path's own origin --; the HBM step subtracts per
// test instead, because the regeneration kernel's Trav::o IS the path's origin (one copy, read back by the shading event).
TRG_DEV void trav_begin(const SceneView &sc, Trav &tv, V3 o, V3 d, float tmax, uint32_t rmask, int sp0, uint32_t node_base = 0u, bool rel = false) {
    // Reciprocal direction with zero components pushed to +-1e-30: the slab products stay finite (no
    // inf - inf = NaN whose fmin/fmax would pick the wrong endpoint), and a ray that moves 1e-30 per
    // unit t along an axis is parallel to the slab for every practical purpose.
    // |d| < 1e-30 ? copysign(1e-30, d) : d, written on the bit patterns (the magnitudes of non-negative floats order like
    // unsigned integers) so that every constant is a VOP2 literal instead of a VGPR the inner loops would have to carry
    const float dx = clamp_away_from_zero(d.x), dy = clamp_away_from_zero(d.y), dz = clamp_away_from_zero(d.z);
    tv.o = (kTriPlanes && rel) ? o - sc.center : o; tv.d = d;
    tv.idx = rcp_fast(dx); tv.idy = rcp_fast(dy); tv.idz = rcp_fast(dz);
    tv.oix = o.x * tv.idx; tv.oiy = o.y * tv.idy; tv.oiz = o.z * tv.idz;
    // the plane test of an LDS-resident scene reads the mask off the u16 per record, (original index << 2) | (material id & 3): only the two low bits
    // of a ray's mask may take part, or they would match index bits.  (The renderers' rays carry 3 or 1; trg_trace sends rays whose mask has
    // higher bits through the HBM records, which keep the whole material id: trg_capi.cpp.)
    tv.best = tmax; tv.rmask = (kTriPlanes && rel) ? (rmask & 3u) : rmask;
    tv.hit.t = -1.0f; tv.hit.prim = -1; tv.hit.u = 0.0f; tv.hit.v = 0.0f;
    tv.found = false;
    tv.node = 0; tv.sp = sp0;  // sp0 = stk.first(): the empty stack
#if TRG_WIDE8
    if (!rel) {   // (HBM traversal) the root as the one inner child, in slot 0, of a virtual parent: popped by the first step
        const uint32_t octinv = ((__float_as_uint(dx) >> 31) | ((__float_as_uint(dy) >> 31) << 1) | ((__float_as_uint(dz) >> 31) << 2)) ^ 7u;
        tv.g_base = 0u; tv.g_bits = (1u << octinv) | (1u << 8); tv.t_base = 0u; tv.t_bits = 0u;
    }
#endif
    // sign-ordered LDS nodes: LDS address of the slab copy to read per axis, for node 0 (node_base = LDS address of the node array; layout:
    // trav_node_step_signed): the [X | Y] block of the sign pair at +0 / +32 / +64 / +96, Z+ Z- at +128 / +160
    tv.sx = (int)(node_base + (((__float_as_uint(dx) >> 31) | ((__float_as_uint(dy) >> 31) << 1)) << 5));
    tv.sz = (int)(node_base + 128u + ((__float_as_uint(dz) >> 31) << 5));
}

// One inner-node step of an LDS-resident scene: test both child boxes of a SIGN-ORDERED BVH2 node, descend into the nearer hit child,
// keep the other on the stack, or pop -- selects and two LDS stack accesses instead of a four-way branch (the branchy form spent more
// scalar instructions on exec-mask bookkeeping than vector instructions on the boxes).  A 208-byte node keeps, per axis,
// the four slab planes of its two children twice: as (lo_a, hi_a, lo_b, hi_b) and as (hi_a, lo_a, hi_b, lo_b).  A
// lane reads the copy that matches the sign of its ray direction (an LDS address fixed per ray, trav_begin), so
// x = near plane, y = far plane without the 12 min/max that order them -- the values are the same floats the
// min/max would pick (the slab product is monotonic in the plane), so hits are unchanged.  The X and Y copies of one
// sign PAIR sit side by side (four 32-byte blocks), and the child pair sits behind BOTH Z copies, so a node costs two
// address adds: one for X and Y, one for Z and the children.  Children are byte offsets of nodes (>= 0) or leaf codes
// (< 0; the record index in them is pre-multiplied by 3 = counted in 16-byte units); 208 = 52 banks, so random nodes
// spread over 16 bank groups.
constexpr int kSignedNodeBytes = 208;
// the 48-byte records of an LDS-resident scene from record `first` on: base + first * 48 as ONE v_mad_u32_u24 (a leaf code holds the record index)
typedef __attribute__((address_space(3))) v4f lds_v4f_t;
TRG_DEV float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool COUNT, int BLOCK, typename STK>
TRG_DEV void trav_node_step_signed(const SceneView &sc, Trav &tv, STK stk, Counters &cnt) {
    if (COUNT) { cnt.nodes++; if (mbcnt64(__ballot(1)) == 0) cnt.wnodes++; }
    const uint32_t az_addr = (uint32_t)(tv.node + tv.sz), axy_addr = (uint32_t)(tv.node + tv.sx);
    const v4f X = *(const lds_v4f_t *)(uintptr_t)axy_addr;
    const v4f Y = *(const lds_v4f_t *)(uintptr_t)(axy_addr + 16u);
    const v4f Z = *(const lds_v4f_t *)(uintptr_t)az_addr;
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2i_t lds_v2i_t;
    const v2i_t cc = *(const lds_v2i_t *)(uintptr_t)(az_addr + 16u);   // the child pair as ONE 8-byte read (2 LDS cycles; as ds_read2_b32: 4)
    const int c0 = cc.x, c1 = cc.y;
    const float anx = X.x * tv.idx - tv.oix, afx = X.y * tv.idx - tv.oix, bnx = X.z * tv.idx - tv.oix, bfx = X.w * tv.idx - tv.oix;
    const float any_ = Y.x * tv.idy - tv.oiy, afy = Y.y * tv.idy - tv.oiy, bny = Y.z * tv.idy - tv.oiy, bfy = Y.w * tv.idy - tv.oiy;
    const float anz = Z.x * tv.idz - tv.oiz, afz = Z.y * tv.idz - tv.oiz, bnz = Z.z * tv.idz - tv.oiz, bfz = Z.w * tv.idz - tv.oiz;
    const float amin = fmaxf(fmaxf(anx, any_), fmaxf(anz, 0.0f));
    // min_raw: a bare v_min_f32.  fminf would first canonicalise tv.best (a loop-carried value the compiler cannot
    // prove quiet) with one more instruction per node; the slab products are never NaN (trav_begin) and best is a
    // finite distance or +inf, so the plain instruction is the same function here.
    const float amax = fminf(fminf(afx, afy), min_raw(afz, tv.best));
    const float bmin = fmaxf(fmaxf(bnx, bny), fmaxf(bnz, 0.0f));
    const float bmax = fminf(fminf(bfx, bfy), min_raw(bfz, tv.best));
    const bool ha = amin <= amax, hb = bmin <= bmax;
    const bool both = ha && hb, none = !(ha || hb);
    const bool first1 = hb && (!ha || bmin < amin);
    const int nearc = first1 ? c1 : c0, farc = first1 ? c0 : c1;
    // the slot above the top is scratch (the stack holds depth + 2 levels, a near-first BVH2 walk keeps at most one
    // pending entry per level): write the far child always and keep it only if both children are entered -- no branch (-1 %)
    stk.push(tv.sp, farc);
    int sp = tv.sp + (both ? STK::unit : 0);
    int next = nearc;
    if (none) {
        sp -= STK::unit;
        next = stk.pop(sp);  // the sentinel at level 0 when nothing is pending
    }
    tv.node = next; tv.sp = sp;
}

// One ray/triangle test folded into the traversal state; returns true when an any-hit query is satisfied.
// REC (HBM scenes): the hit keeps the leaf RECORD index `rec` instead of the original primitive index.  Distance ties still go to the
// lower ORIGINAL index (the intersection contract): only then -- two triangles at bit-equal distance, i.e. duplicates or a ray
// through a shared edge -- is the held triangle's original index read back from its record (recs[held * kRecV4].w).
template <bool COUNT, bool REC = false>
TRG_DEV bool trav_tri_math(const v4f a, const v4f b, const v4f c, Trav &tv, bool any, Counters &cnt, uint32_t rec = 0u, const v4f *recs = nullptr) {
    const bool masked_in = (((uint32_t)__float_as_int(b.w)) & tv.rmask) != 0u;
    if (COUNT) { if (masked_in) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
    float t, u, v;
    // t is tested against `best` instead of the ray's own maxDistance: best <= maxDistance, and a triangle between the two could not
    // change anything (something nearer is already held, so `found` is set and `take` would be false) -- one register less per ray
    const bool ok = tri_test(a, b, c, tv.o, tv.d, tv.best, t, u, v) && masked_in;
    const int prim = __float_as_int(a.w);
    bool take;
    if (REC) {
        const bool closer = any || !tv.found || t < tv.best;
        take = ok && closer;
        if (ok && !closer && t == tv.best) take = prim < fat_prim(recs, (uint32_t)tv.hit.prim);
    } else {
        take = ok && (any || !tv.found || t < tv.best || (t == tv.best && prim < tv.hit.prim));
    }
    tv.found = tv.found || ok;
    tv.best = take ? t : tv.best;   // (an any-hit query ends with this triangle: its distance may overwrite the limit)
    tv.hit.prim = take ? (REC ? (int)rec : prim) : tv.hit.prim;
    tv.hit.u = take ? u : tv.hit.u;
    tv.hit.v = take ? v : tv.hit.v;
    return any && ok;
}

// the plane form of the same fold (LDS-resident scenes, shipped build): `meta` = (original index << 2) | (material id & 3)
template <bool COUNT>
TRG_DEV bool trav_tri_planes(const v4f a, const v4f b, const v4f c, uint32_t meta, Trav &tv, bool any, Counters &cnt) {
    const bool masked_in = (meta & tv.rmask) != 0u;   // (tv.rmask has bits 0..1 only here: trav_begin)
    if (COUNT) { if (masked_in) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
    float t, u, v;
    const bool ok = tri_test_planes(a, b, c, tv.o, tv.d, tv.best, t, u, v) && masked_in;
    const int prim = (int)(meta >> 2);
    const bool take = ok && (any || !tv.found || t < tv.best || (t == tv.best && prim < tv.hit.prim));
    tv.found = tv.found || ok;
    tv.best = take ? t : tv.best;
    tv.hit.prim = take ? prim : tv.hit.prim;
    tv.hit.u = take ? u : tv.hit.u;
    tv.hit.v = take ? v : tv.hit.v;
    return any && ok;
}
// a QUAD leaf of an LDS-resident scene: one parallelogram test for the two records starting at `rec`
template <bool COUNT>
TRG_DEV bool trav_quad_planes(const SceneView &sc, const v4f *tr, uint32_t rec, Trav &tv, bool any, Counters &cnt) {
    const uint32_t meta0 = sc.meta[rec];
    const bool masked_in = (meta0 & tv.rmask) != 0u;   // (the two triangles of a quad have one material; tv.rmask has bits 0..1 only: trav_begin)
    if (COUNT) { if (masked_in) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
    float t, u, v;
    bool second;
    const bool ok = tri_test_planes_quad(tr[0], tr[1], tr[2], tv.o, tv.d, tv.best, true, t, u, v, second) && masked_in;
    const int prim = (int)((second ? (uint32_t)sc.meta[rec + 1u] : meta0) >> 2);
    const bool take = ok && (any || !tv.found || t < tv.best || (t == tv.best && prim < tv.hit.prim));
    tv.found = tv.found || ok;
    tv.best = take ? t : tv.best;
    tv.hit.prim = take ? prim : tv.hit.prim;
    tv.hit.u = take ? u : tv.hit.u;
    tv.hit.v = take ? v : tv.hit.v;
    return any && ok;
}
// a BOX leaf of an LDS-resident scene (shipped build; bvh_build.h kLeafBox, trg_capi.cpp): `tr` = its twelve plane records -- six quads, X then Y
// each.  A quad's test reads its X record only, so the Y slots are free: record 1 holds the box, rows (a_k, d_k) with l_k = a_k . o + d_k the
// ray origin in the box's own frame (inside <=> |l_k| <= 1), record 3 one word per face f = 2 k + (l_k > 0): (X-record offset << 16) | its byte
// offset.  One slab test in that frame; the face the ray enters by (leaves by, from inside: the triangles are two-sided) names the quad, and that
// quad's own rows 1, 2 give (s, t) at the hit point exactly as its quad test would -- hence the triangle and the weights the reference's buffers know.
template <bool COUNT>
TRG_DEV bool trav_box_planes(const SceneView &sc, const v4f *tr, uint32_t first, Trav &tv, bool any, Counters &cnt) {
    const uint32_t meta0 = sc.meta[first];
    const bool masked_in = (meta0 & tv.rmask) != 0u;   // (one material for the twelve: bvh_build.cpp box_group)
    if (COUNT) { if (masked_in) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
    const v4f b0 = tr[3], b1 = tr[4], b2 = tr[5];
    const V3 o = tv.o, d = tv.d;
    const float lox = b0.x * o.x + (b0.y * o.y + (b0.z * o.z + b0.w)), ldx = b0.x * d.x + (b0.y * d.y + b0.z * d.z);
    const float loy = b1.x * o.x + (b1.y * o.y + (b1.z * o.z + b1.w)), ldy = b1.x * d.x + (b1.y * d.y + b1.z * d.z);
    const float loz = b2.x * o.x + (b2.y * o.y + (b2.z * o.z + b2.w)), ldz = b2.x * d.x + (b2.y * d.y + b2.z * d.z);
    const float ix = rcp_fast(ldx), iy = rcp_fast(ldy), iz = rcp_fast(ldz);
    const float mx = -lox * ix, my = -loy * iy, mz = -loz * iz;              // the ray meets the planes l_k = -1, +1 at m_k -+ |1 / ld_k|
    const float ax = fabsf(ix), ay = fabsf(iy), az = fabsf(iz);
    const float tnear = fmaxf(fmaxf(mx - ax, my - ay), mz - az), tfar = fminf(fminf(mx + ax, my + ay), mz + az);
    const float t = tnear < 0.0f ? tfar : tnear;                             // (from inside: the triangles are two-sided, the ray meets the face it leaves by)
    const bool ok = (tnear <= tfar) && (t >= 0.0f) && (t <= tv.best) && masked_in;
    // only the slab test inside the traversal loop: an accepted hit keeps the distance and -(first + 2) -- "a box, triangle and weights still to be
    // named" -- and lds_box_hit_resolve() names them once, for the hit that survived, when the traversal is over (an any-hit query never needs them)
    const bool take = ok && (any || !tv.found || t < tv.best);
    tv.found = tv.found || ok;
    tv.best = take ? t : tv.best;
    tv.hit.prim = take ? -(int)(first + 2u) : tv.hit.prim;
    return any && ok;
}
TRG_DEV const v4f *lds_records(const SceneView &sc, uint32_t first);
// the triangle and the weights of a hit that is still a box (h.prim <= -2): the hit point in the box's frame, the face = the axis along which it lies
// farthest out, that face's own quad planes for (s, t) -- exactly what its quad test would have computed there
TRG_DEV void lds_box_hit_resolve(const SceneView &sc, const Trav &tv, Hit &h) {
    if (!(kTriPlanes && TRG_BOX_LEAVES) || h.prim > -2) return;
    const uint32_t first = (uint32_t)(-h.prim - 2);
    const v4f *tr = lds_records(sc, first);
    const v4f b0 = tr[3], b1 = tr[4], b2 = tr[5];
    const V3 o = tv.o, d = tv.d;
    const V3 P = mk(o.x + h.t * d.x, o.y + h.t * d.y, o.z + h.t * d.z);
    const float lx = b0.x * P.x + (b0.y * P.y + (b0.z * P.z + b0.w)), ly = b1.x * P.x + (b1.y * P.y + (b1.z * P.z + b1.w)), lz = b2.x * P.x + (b2.y * P.y + (b2.z * P.z + b2.w));
    const float fx = fabsf(lx), fy = fabsf(ly), fz = fabsf(lz);
    const bool isz = fz >= fx && fz >= fy, isy = !isz && fy >= fx, isx = !isz && !isy;
    const float lk = isx ? lx : isy ? ly : lz;
    const uint32_t f = (isx ? 0u : isy ? 2u : 4u) + (lk > 0.0f ? 1u : 0u);
    const uint32_t fw = reinterpret_cast<const uint32_t *>(tr + 9)[f];
    const v4f *q = reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(tr) + (fw & 0xFFFFu));
    const v4f q1 = q[1], q2 = q[2];
    const float s0 = q1.x * P.x + (q1.y * P.y + (q1.z * P.z + q1.w));
    const float t0 = q2.x * P.x + (q2.y * P.y + (q2.z * P.z + q2.w));
    const bool second = s0 < t0;
    h.u = second ? s0 : s0 - t0;
    h.v = second ? t0 - s0 : t0;
    h.prim = (int)((uint32_t)sc.meta[first + (fw >> 16) + (second ? 1u : 0u)] >> 2);
}
// the original index a held HBM hit stands for in the distance-tie rule: its record's, or -- a hit that is still a BOX (index >= n_rec, trav_box_rec) --
// that of the box's first record: a box's triangles are consecutive, so any of them orders it against everything outside the box
TRG_DEV int held_prim_index(const v4f *recs, uint32_t n_rec, int held) {
    const uint32_t r = (uint32_t)held;
    return fat_prim(recs, (kBoxHbm && r >= n_rec) ? (uint32_t)__float_as_int(recs[(size_t)r * kRecV4 + 3].x) : r);
}
// ... and on a leaf RECORD of an HBM-resident scene (the hit keeps the record index; ties go to the lower original index, read back from the
// held record only then): mask and prim are floats 13 and 12 of the record (TRG_REC_META_FIRST; its last two words before)
template <bool COUNT>
TRG_DEV bool trav_tri_planes_rec(const v4f a, const v4f b, const v4f c, uint32_t mask, int prim, Trav &tv, bool any, Counters &cnt, uint32_t rec, const v4f *recs, V3 center, bool quad, uint32_t n_rec) {
    const bool masked_in = (mask & tv.rmask) != 0u;
    if (COUNT) { if (masked_in) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
    float t, u, v;
    bool second;   // quad leaf: the hit lies in the SECOND triangle of the parallelogram (the next record)
    const bool ok = tri_test_planes_quad(a, b, c, tv.o - center, tv.d, tv.best, quad, t, u, v, second) && masked_in;   // (Trav::o is the path's origin here: trav_begin)
    rec += second ? 1u : 0u;
    const bool closer = any || !tv.found || t < tv.best;
    bool take = ok && closer;
    if (ok && !closer && t == tv.best) take = (second ? fat_prim(recs, rec) : prim) < held_prim_index(recs, n_rec, tv.hit.prim);
    tv.found = tv.found || ok;
    tv.best = take ? t : tv.best;
    tv.hit.prim = take ? (int)rec : tv.hit.prim;
    tv.hit.u = take ? u : tv.hit.u;
    tv.hit.v = take ? v : tv.hit.v;
    return any && ok;
}
// A BOX leaf of a scene traversed from HBM (shipped build; trg_kernels.h SceneDesc::off_boxrec): q0..q2 = the box's frame, q3 = (first leaf record,
// material id, face table low / high).  The same slab test as trav_box_planes -- and nothing else inside the traversal loop: an accepted hit keeps
// the distance and the index of the BOX record (`self`: it sits behind the leaf records, so any index >= SceneView::n_rec says "a box, face and
// weights still to be named"); box_hit_resolve() names them once, for the hit that survived, when the traversal is over (any-hit queries never need
// them).
template <bool COUNT>
TRG_DEV bool trav_box_rec(const v4f b0, const v4f b1, const v4f b2, const v4f b3, Trav &tv, bool any, Counters &cnt, uint32_t self, V3 center, const v4f *recs, uint32_t n_rec) {
    const uint32_t mask = (uint32_t)__float_as_int(b3.y);
    const bool masked_in = (mask & tv.rmask) != 0u;
    if (COUNT) { if (masked_in) cnt.tris++; if (mbcnt64(__ballot(1)) == 0) cnt.wtris++; }
    const V3 o = tv.o - center, d = tv.d;
    const float lox = b0.x * o.x + (b0.y * o.y + (b0.z * o.z + b0.w)), ldx = b0.x * d.x + (b0.y * d.y + b0.z * d.z);
    const float loy = b1.x * o.x + (b1.y * o.y + (b1.z * o.z + b1.w)), ldy = b1.x * d.x + (b1.y * d.y + b1.z * d.z);
    const float loz = b2.x * o.x + (b2.y * o.y + (b2.z * o.z + b2.w)), ldz = b2.x * d.x + (b2.y * d.y + b2.z * d.z);
    const float ix = rcp_fast(ldx), iy = rcp_fast(ldy), iz = rcp_fast(ldz);
    const float mx = -lox * ix, my = -loy * iy, mz = -loz * iz;              // the ray meets the planes l_k = -1, +1 at m_k -+ |1 / ld_k|
    const float ax = fabsf(ix), ay = fabsf(iy), az = fabsf(iz);
    const float tnear = fmaxf(fmaxf(mx - ax, my - ay), mz - az), tfar = fminf(fminf(mx + ax, my + ay), mz + az);
    const float t = tnear < 0.0f ? tfar : tnear;                             // (from inside: the triangles are two-sided, the ray meets the face it leaves by)
    const bool ok = (tnear <= tfar) && (t >= 0.0f) && (t <= tv.best) && masked_in;
    const bool closer = any || !tv.found || t < tv.best;
    bool take = ok && closer;
    // the distance-tie rule (lower original index), as far as it can be told before the triangle is named: by the box's first record (read only at
    // bit-equal distances -- an exact duplicate of a quad, two cubes face to face)
    if (ok && !closer && t == tv.best) take = fat_prim(recs, (uint32_t)__float_as_int(b3.x)) < held_prim_index(recs, n_rec, tv.hit.prim);
    tv.found = tv.found || ok;
    tv.best = take ? t : tv.best;
    tv.hit.prim = take ? (int)self : tv.hit.prim;
    return any && ok;
}
// ... and, after the traversal, the triangle and the weights of a hit that is still a box (h.prim >= sc.n_rec): the hit point in the box's frame,
// the face = the axis along which it lies farthest out (|l_k| = 1 there, less on the other two; a lone quad dressed as a box of no thickness
// says so in bit 31 of the table's high word: its l_2 is rounding noise), then (s, t) from the two OTHER coordinates and three bits per face.
TRG_DEV void box_hit_resolve(const SceneView &sc, const Trav &tv, Hit &h) {
    if (!kBoxHbm || h.prim < (int)sc.n_rec) return;
    const v4f *br = sc.tris + (size_t)(uint32_t)h.prim * kRecV4;
    const v4f b0 = br[0], b1 = br[1], b2 = br[2], b3 = br[3];
    const V3 o = tv.o - sc.center, d = tv.d;
    const V3 P = mk(o.x + h.t * d.x, o.y + h.t * d.y, o.z + h.t * d.z);
    const float lx = b0.x * P.x + (b0.y * P.y + (b0.z * P.z + b0.w)), ly = b1.x * P.x + (b1.y * P.y + (b1.z * P.z + b1.w)), lz = b2.x * P.x + (b2.y * P.y + (b2.z * P.z + b2.w));
    const float fx = fabsf(lx), fy = fabsf(ly), fz = fabsf(lz);
    const uint32_t hiw = (uint32_t)__float_as_int(b3.w);
    const bool isz = (hiw >> 31) != 0u || (fz >= fx && fz >= fy), isy = !isz && fy >= fx, isx = !isz && !isy;   // (bit 31: a lone quad -- its face is axis 2 whatever the noise says)
    const float lk = isx ? lx : isy ? ly : lz;
    const uint32_t f = (isx ? 0u : isy ? 2u : 4u) + (lk > 0.0f ? 1u : 0u);
    const uint32_t tw = (f >= 4u ? hiw : (uint32_t)__float_as_int(b3.z)) >> (7u * (f & 3u));
    const float li = isx ? ly : lx, lj = (isx || isy) ? lz : ly;               // the two other coordinates, i < j
    const bool swap = (tw & 16u) != 0u;
    const float ls = swap ? lj : li, lt = swap ? li : lj;
    const float s0 = ls * ((tw & 32u) ? -0.5f : 0.5f) + 0.5f, t0 = lt * ((tw & 64u) ? -0.5f : 0.5f) + 0.5f;
    const bool second = s0 < t0;
    h.u = second ? s0 : s0 - t0;
    h.v = second ? t0 - s0 : t0;
    h.prim = (int)((uint32_t)__float_as_int(b3.x) + (tw & 15u) + (second ? 1u : 0u));
}
// triangle `k` of the records starting at `tr` (LDS-resident scene): whichever test the build uses
template <bool COUNT>
TRG_DEV bool trav_tri_lds(const SceneView &sc, const v4f *tr, uint32_t rec, Trav &tv, bool any, Counters &cnt) {
    if (kTriPlanes) return trav_tri_planes<COUNT>(tr[0], tr[1], tr[2], sc.meta[rec], tv, any, cnt);
    return trav_tri_math<COUNT>(tr[0], tr[1], tr[2], tv, any, cnt);
}

// the hit record of a finished traversal
TRG_DEV Hit trav_hit(const Trav &tv) {
    Hit h = tv.hit;
    h.t = tv.found ? tv.best : -1.0f;
    return h;
}

// One leaf: test its 1..8 triangles.  ~node = (first << 3) | (count - 1).  Pops the next node (the sentinel kNodeDone when
// nothing is pending) and returns true when an any-hit query is satisfied -- the caller then stops whatever was popped.
TRG_DEV const v4f *lds_records(const SceneView &sc, uint32_t first) {
    return reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(sc.tris) + __umul24(first, 48u));
}
template <bool COUNT, int BLOCK, typename STK>
TRG_DEV bool trav_leaf_step(const SceneView &sc, Trav &tv, bool any, STK stk, Counters &cnt) {
    const uint32_t code = (uint32_t)~tv.node;
    const bool quad = (code & 7u) == kLeafQuad;   // the two triangles of a parallelogram (bvh_build.h): ONE plane test in the shipped build
    const bool box = TRG_BOX_LEAVES && (code & 7u) == kLeafBox;   // the twelve of a parallelepiped: ONE slab test in the shipped build
    const uint32_t first = code >> 3, count = quad ? 2u : box ? 12u : (code & 7u) + 1u;
    bool stop = false;
    {   // leaves of the host builder hold a quad, or one or two triangles: those without a loop (-2 %); more only from other builders
        const v4f *tr = lds_records(sc, first);
        if (kTriPlanes && TRG_BOX_LEAVES && box) {
            stop = trav_box_planes<COUNT>(sc, tr, first, tv, any, cnt);
        } else if (kTriPlanes && quad) {
            stop = trav_quad_planes<COUNT>(sc, tr, first, tv, any, cnt);
        } else {
            stop = trav_tri_lds<COUNT>(sc, tr, first, tv, any, cnt);
            if (!stop && count > 1u) stop = trav_tri_lds<COUNT>(sc, tr + 3, first + 1u, tv, any, cnt);
            for (uint32_t k = 2; k < count && !stop; ++k) stop = trav_tri_lds<COUNT>(sc, tr + k * 3, first + k, tv, any, cnt);
        }
    }
    const int sp = tv.sp - STK::unit;
    tv.node = stk.pop(sp);
    tv.sp = sp;
    return stop;
}

// ---- 4-wide nodes (scenes in HBM): 128-byte node, four child boxes in SoA form (bvh_build.h) ----
#define TRG_CSWAP(ta, ca, tb, cb)                         \
    {                                                     \
        const bool sw_ = (tb) < (ta);                     \
        const float tt_ = sw_ ? (tb) : (ta);              \
        (tb) = sw_ ? (ta) : (tb);                         \
        (ta) = tt_;                                       \
        const int cc_ = sw_ ? (cb) : (ca);                \
        (cb) = sw_ ? (ca) : (cb);                         \
        (ca) = cc_;                                       \
    }

// Common tail of the 4-wide node steps: t_k = entry distance of child k or +inf when the ray does not enter it.  Sorts the four
// (distance, child) pairs nearest first (5-comparator network; misses sink to the end), pushes the farther hits, continues with
// the nearest or pops.
template <int BLOCK, typename STK>
TRG_DEV void wide_select(float t0, float t1, float t2, float t3, int c0, int c1, int c2, int c3, Trav &tv, STK stk) {
    TRG_CSWAP(t0, c0, t1, c1) TRG_CSWAP(t2, c2, t3, c3) TRG_CSWAP(t0, c0, t2, c2) TRG_CSWAP(t1, c1, t3, c3) TRG_CSWAP(t1, c1, t2, c2)
    const bool h4 = t3 < INFINITY, h3 = t2 < INFINITY, h2 = t1 < INFINITY, h0 = !(t0 < INFINITY);  // sorted: t_k finite <=> more than k hits
    int sp = tv.sp;
    if (stk.room(sp, 3)) {
        // the slots above the top are scratch: store the three candidates where they would go and keep what was entered -- no
        // branch, no per-push bounds test (a store at an unchanged sp is overwritten by the next one: LDS stores of a wave stay in order)
        stk.put(sp, c3);
        sp += h4 ? STK::unit : 0;
        stk.put(sp, c2);
        sp += h3 ? STK::unit : 0;
        stk.put(sp, c1);
        sp += h2 ? STK::unit : 0;
    } else {   // within three entries of the end of the LDS levels: entry by entry, the deeper ones go to the scratch in memory
        if (h4) stk.push(sp, c3);
        sp += h4 ? STK::unit : 0;
        if (h3) stk.push(sp, c2);
        sp += h3 ? STK::unit : 0;
        if (h2) stk.push(sp, c1);
        sp += h2 ? STK::unit : 0;
    }
    int next = c0;
    if (h0) {
        sp -= STK::unit;
        next = stk.pop(sp);  // the sentinel at level 0 when nothing is pending
    }
    tv.node = next; tv.sp = sp;
}

// One 4-wide step on a QUANTISED node (q4node.h): 64 bytes = four 16-byte loads.  plane = origin + q * scale, so
// t = (plane - o) / d = q * (scale / d) + (origin / d - o / d): two ray-dependent constants per axis (A, B), then one
// v_cvt_f32_ubyteN + one fma per plane.  The near / far plane dwords are picked by the sign of the direction (one
// select per axis for all four children), which also removes the per-child min/max ordering.
template <bool COUNT, int BLOCK, typename STK>
TRG_DEV void trav_node4_math(const v4f q0, const v4f q1, const v4f q2, const v4f ch, Trav &tv, STK stk, Counters &cnt) {
    if (COUNT) { cnt.nodes += 2; if (mbcnt64(__ballot(1)) == 0) cnt.wnodes++; }  // 4 boxes = 2 BVH2-equivalent fetches
    int c0 = __float_as_int(ch.x), c1 = __float_as_int(ch.y), c2 = __float_as_int(ch.z), c3 = __float_as_int(ch.w);
    const float ax = q0.w * tv.idx, ay = q2.z * tv.idy, az = q2.w * tv.idz;
    const float bx = q0.x * tv.idx - tv.oix, by = q0.y * tv.idy - tv.oiy, bz = q0.z * tv.idz - tv.oiz;
    // direction signs off the reciprocals (1 / d keeps the sign of d): nothing to keep in registers for them
    const bool negx = (__float_as_uint(tv.idx) >> 31) != 0u, negy = (__float_as_uint(tv.idy) >> 31) != 0u, negz = (__float_as_uint(tv.idz) >> 31) != 0u;
    const uint32_t lox = __float_as_uint(q1.x), hix = __float_as_uint(q1.y), loy = __float_as_uint(q1.z), hiy = __float_as_uint(q1.w);
    const uint32_t loz = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
    const uint32_t nx = negx ? hix : lox, fx = negx ? lox : hix;
    const uint32_t ny = negy ? hiy : loy, fy = negy ? loy : hiy;
    const uint32_t nz = negz ? hiz : loz, fz = negz ? loz : hiz;
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tnx = (float)((nx >> (8 * k)) & 255u) * ax + bx, tfx = (float)((fx >> (8 * k)) & 255u) * ax + bx;
        const float tny = (float)((ny >> (8 * k)) & 255u) * ay + by, tfy = (float)((fy >> (8 * k)) & 255u) * ay + by;
        const float tnz = (float)((nz >> (8 * k)) & 255u) * az + bz, tfz = (float)((fz >> (8 * k)) & 255u) * az + bz;
        const float tmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tmax = fminf(fminf(tfx, tfy), min_raw(tfz, tv.best));   // (min_raw: trav_node_step_signed)
        // +inf = not entered.  An unused slot decodes to the inverted box (255, 0) and can never pass, so the child word
        // needs no check of its own (-4 % on C4)
        t[k] = (tmin <= tmax) ? tmin : INFINITY;
    }
    wide_select<BLOCK>(t[0], t[1], t[2], t[3], c0, c1, c2, c3, tv, stk);
}

// one unit of work per lane per iteration on the 4-wide tree: a quantised node (four 16-byte loads = 64 bytes)
// or one triangle of the current leaf (the first three 16-byte rows of its 128-byte record)
template <bool COUNT, int BLOCK, typename STK>
TRG_DEV void trav_step_wide(const SceneView &sc, Trav &tv, bool any, STK stk, Counters &cnt) {
    const bool inner = tv.node >= 0;
    const uint32_t code = (uint32_t)~tv.node;
    const uint32_t first = code >> 3, left = code & 7u;
    // one wave-uniform base + a 32-bit byte offset per lane (the scene blob is below 4 GiB and the records follow the nodes in it):
    // the loads take the SGPR-base form, no 64-bit address arithmetic per lane
    const bool boxleaf = kBoxHbm && !inner && left == kLeafBox;   // (shipped build: ONE leaf for a parallelepiped's twelve triangles; its 64-byte record -- addressed like a
                                                                  //  leaf record behind the leaf records -- is all the test reads)
    const uint32_t off = inner ? (uint32_t)tv.node * 64u : (code & ~7u) * (uint32_t)(kRecV4 * 16 / 8) + sc.rec_delta;
    const v4f *ptr = reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(sc.nodes) + off);
    const v4f q0 = ptr[0], q1 = ptr[1], q2 = ptr[2];
    // the fourth load: the children of a node, or -- plane records -- the last row of the leaf record (its index and mask); one instruction for both kinds
    v4f q3;
    if (kRecMetaFirst) q3 = ptr[3];          // (the record's first 64 bytes hold all a test reads: one line half, like a node)
    else if (kRecPlanes) q3 = ptr[inner ? 3 : 7];
    else if (inner) q3 = ptr[3];
    if (inner) {
        trav_node4_math<COUNT, BLOCK>(q0, q1, q2, q3, tv, stk, cnt);
    } else if (kBoxHbm && boxleaf) {
        const bool stop = trav_box_rec<COUNT>(q0, q1, q2, q3, tv, any, cnt, first, sc.center, sc.tris, sc.n_rec);
        const int sp = tv.sp - (stop ? 0 : STK::unit);
        const int popped = stk.pop(sp);
        tv.node = stop ? kNodeDone : popped;
        tv.sp = sp;
    } else {
        // a QUAD leaf (count field 7, bvh_build.h): the shipped build decides both triangles with one parallelogram test; the strict build
        // tests record `first`, then advances to the single-triangle code of record first + 1
        const bool quad = left == kLeafQuad;
        const bool stop = kRecPlanes ? trav_tri_planes_rec<COUNT>(q0, q1, q2, (uint32_t)__float_as_int(kRecMetaFirst ? q3.y : q3.w), __float_as_int(kRecMetaFirst ? q3.x : q3.z), tv, any, cnt, first, sc.tris, sc.center, quad, sc.n_rec)
                                     : trav_tri_math<COUNT, true>(q0, q1, q2, tv, any, cnt, first, sc.tris);
        const bool more = kRecPlanes ? (left != 0u && !quad) : (left != 0u);
        const bool do_pop = !stop && !more;
        const int sp = tv.sp - (do_pop ? STK::unit : 0);
        const int popped = stk.pop(sp);  // the sentinel at level 0 when nothing is pending
        // the next triangle of the leaf: ~(((first + 1) << 3) | (left - 1)) = node - 7 for left >= 1; after the first triangle of a quad
        // (strict build) ~(((first + 1) << 3) | 0) = node - 1
        const int advanced = tv.node - ((!kRecPlanes && quad) ? 1 : 7);
        tv.node = stop ? kNodeDone : (more ? advanced : popped);
        tv.sp = sp;
    }
}

#if TRG_WIDE8
#include "../../experiments/trg_wide8.inc.h"   // trav_step_wide8: the compressed 8-wide tree (measured slower: profiles/r05/c4_wide8_experiment.md)
#endif  // TRG_WIDE8
// the unit of work of an HBM traversal in this build
template <bool COUNT, int BLOCK, typename STK>
TRG_DEV void trav_step_hbm(const SceneView &sc, Trav &tv, bool any, STK stk, Counters &cnt) {
#if TRG_WIDE8
    static_assert(!kRecPlanes || kRecMetaFirst, "TRG_WIDE8 reads a plane record's index word and mask from floats 12, 13");
    trav_step_wide8<COUNT, BLOCK>(sc, tv, any, stk, cnt);
#else
    trav_step_wide<COUNT, BLOCK>(sc, tv, any, stk, cnt);
#endif
}

// Traversal schedules.  LDS-resident scenes: while-while on the sign-ordered BVH2 nodes.  HBM-resident scenes (UNIFIED): the unified step on
// the quantised 4-wide tree -- one unit of work per lane per iteration, a node or one triangle, off one group of loads.  (Measured and
// dropped, NOTEBOOK.md: a plain BVH2, a sign-ordered 4-wide float tree, an octant-threaded stackless walk and half-precision nodes in LDS;
// the unified step on BVH2 nodes and a block-kind vote per iteration in HBM.)

// LDS address of the node array when the traversal reads sign-ordered LDS nodes (UNIFIED = false selects the LDS schedule)
template <bool UNIFIED>
TRG_DEV uint32_t lds_node_base(const SceneView &sc) {
    return !UNIFIED ? (uint32_t)(uintptr_t)(const lds_v4f_t *)sc.nodes : 0u;
}

#if TRG_EXPERIMENTS
#include "../../experiments/trg_flat.inc.h"   // traverse_flat: tiny scenes without a tree (measured: C2 +7 % time, C3 -5 %: profiles/r05/c2_flat_list_experiment.md)
#endif

// Nearest-hit (ANY=false) or any-hit (ANY=true) query for one ray per lane.
template <bool ANY, bool COUNT, int BLOCK, bool UNIFIED = false, typename STK>
TRG_DEV bool traverse(const SceneView &sc, V3 o, V3 d, float tmax_ray, uint32_t rmask, Hit &hit, STK stk,
                      Counters &cnt) {
#ifdef TRG_EXP_NOTRAVERSE  // timing-only ablation: a made-up hit, no traversal
    hit.t = 0.5f + 0.001f * (float)(threadIdx.x & 15); hit.prim = ANY ? -1 : (int)((threadIdx.x * 7u + (uint32_t)(o.x * 64.0f)) % 34u); hit.u = 0.3f; hit.v = 0.3f;
    return !ANY;
#endif
#if TRG_EXPERIMENTS
    if (!UNIFIED && kTriPlanes && sc.n_flat != 0u) return traverse_flat<ANY, COUNT>(sc, o, d, tmax_ray, rmask, hit, cnt);   // (wave-uniform: a property of the scene)
#endif
    Trav tv;
    trav_begin(sc, tv, o, d, tmax_ray, rmask, stk.first(), lds_node_base<UNIFIED>(sc), !UNIFIED);
    if (UNIFIED) {
        while (tv.node != kNodeDone) trav_step_hbm<COUNT, BLOCK>(sc, tv, ANY, stk, cnt);
    } else {
        for (;;) {
            while (tv.node >= 0) trav_node_step_signed<COUNT, BLOCK>(sc, tv, stk, cnt);
            if (tv.node == kNodeDone) break;
            if (trav_leaf_step<COUNT, BLOCK>(sc, tv, ANY, stk, cnt)) break;   // any-hit satisfied
            if (tv.node == kNodeDone) break;
        }
    }
    hit = trav_hit(tv);
    if (UNIFIED && !ANY) box_hit_resolve(sc, tv, hit);
    if (!UNIFIED && !ANY) lds_box_hit_resolve(sc, tv, hit);
    return tv.found;
}

// Two rays per lane in ONE traversal loop: the shadow ray of bounce b (any-hit) and the continuation ray of
// bounce b+1 (nearest-hit) both come out of the same shading event and are independent, so a lane walks
// them back to back without waiting for the other lanes in between.  The wavefront then pays
// max over lanes of (len_shadow + len_next) instead of max(len_shadow) + max(len_next) -- the cheapest form of
// decoupling lanes from one another: no queue, no LDS traffic, identical arithmetic per ray.
template <bool COUNT, int BLOCK, bool UNIFIED, typename STK>
TRG_DEV void traverse_pair(const SceneView &sc, V3 org, bool has_shadow, V3 sdir, float smax, bool has_next, V3 ndir,
                           uint32_t nmask, bool &occluded, Hit &nhit, bool &nfound, STK stk, Counters &cnt) {
    occluded = false; nfound = false;
    nhit.t = -1.0f; nhit.prim = -1; nhit.u = 0.0f; nhit.v = 0.0f;
    int phase = has_shadow ? 0 : (has_next ? 1 : 2);
    Trav tv;
    trav_begin(sc, tv, org, phase == 0 ? sdir : ndir, phase == 0 ? smax : INFINITY, phase == 0 ? 1u : nmask, stk.first(), lds_node_base<UNIFIED>(sc), !UNIFIED);
    if (phase == 2) tv.node = kNodeDone;
    while (phase < 2) {
        const bool any = phase == 0;
        if (UNIFIED) {
            trav_step_hbm<COUNT, BLOCK>(sc, tv, any, stk, cnt);
        } else {
            while (tv.node >= 0) trav_node_step_signed<COUNT, BLOCK>(sc, tv, stk, cnt);
            if (tv.node != kNodeDone && trav_leaf_step<COUNT, BLOCK>(sc, tv, any, stk, cnt)) tv.node = kNodeDone;
        }
        if (tv.node == kNodeDone) {
            if (phase == 0) {
                occluded = tv.found;
                phase = has_next ? 1 : 2;
                if (phase == 1) trav_begin(sc, tv, org, ndir, INFINITY, nmask, stk.first(), lds_node_base<UNIFIED>(sc), !UNIFIED);
            } else {
                nhit = trav_hit(tv); nfound = tv.found;
                if (UNIFIED) box_hit_resolve(sc, tv, nhit); else lds_box_hit_resolve(sc, tv, nhit);
                phase = 2;
            }
        }
    }
}

// Raytracing.metal:95-112 with coordinates = (1-u-v, u): weights of vertex 0 and vertex 1
TRG_DEV V3 interp_attr(const float *attr, int prim, float cx, float cy) {
    const float cz = 1.0f - cx - cy;
    const float *p = attr + prim * 9;
    const V3 T0 = mk(p[0], p[1], p[2]), T1 = mk(p[3], p[4], p[5]), T2 = mk(p[6], p[7], p[8]);
    return cx * T0 + cy * T1 + cz * T2;
}

// Albedo texel of a shaded hit (trg_load_textures; project definition, the reference has no textures): the texture coordinates
// are interpolated like every other vertex attribute (Raytracing.metal:95-112), the lookup is nearest-texel with repeat wrap.
TRG_DEV V3 texture_albedo(const trg::TexDesc &tex, int prim, float cx, float cy, V3 vcol) {
    const uint32_t id = tex.ids[prim];
    if (id == 0u) return vcol;
    const float cz = 1.0f - cx - cy;
    const float *p = tex.uv + prim * 6;
    const float u = cx * p[0] + cy * p[2] + cz * p[4], v = cx * p[1] + cy * p[3] + cz * p[5];
    const uint32_t *t = tex.table + (id - 1u) * 4u;
    const uint32_t w = t[1], h = t[2];
    const float fu = u - floorf(u), fv = v - floorf(v);
    uint32_t x = (uint32_t)(fu * (float)w), y = (uint32_t)(fv * (float)h);
    x = x < w ? x : w - 1u; y = y < h ? y : h - 1u;
    const uint32_t texel = tex.texels[t[0] + y * w + x];
    const V3 c = mk(div_fast((float)(texel & 255u), 255.0f), div_fast((float)((texel >> 8) & 255u), 255.0f), div_fast((float)((texel >> 16) & 255u), 255.0f));
    return vcol * c;
}

// The surface a shading event reads at its hit (Raytracing.metal:150-163: material id, interpolated colour and normal).
// FAT = the scene is traversed from HBM and `ref` is the leaf record of the hit: the material id (the mask of row 1) and the five
// rows of attributes are loaded TOGETHER, before anything branches on the material -- one trip to a line the triangle test has
// just touched, instead of material id -> branch -> colours + normals from three more arrays.  !FAT (scene in LDS): `ref` is the
// original primitive index and the reference's three buffers are read as before.  Same arithmetic either way (interp_attr).
template <bool FAT>
struct Surf { uint32_t mat; int ref; v4f r3, r4, r5, r6, r7; };
template <bool FAT>
TRG_DEV Surf<FAT> surf_fetch(const SceneView &sc, int ref) {
    Surf<FAT> s;
    s.ref = ref;
    if (FAT) {
        const v4f *rec = sc.tris + (size_t)(uint32_t)ref * kRecV4;
        s.mat = (uint32_t)__float_as_int(kRecMetaFirst ? rec[3].y : kRecPlanes ? rec[7].w : rec[1].w);
        s.r3 = rec[3]; s.r4 = rec[4]; s.r5 = rec[5]; s.r6 = rec[6]; s.r7 = rec[7];
    } else {
        s.mat = sc.mats[ref];
    }
    return s;
}
// interpolated vertex colour (times the albedo texel, if the scene has textures) and the un-normalised interpolated normal
template <bool FAT>
TRG_DEV void surf_interp(const SceneView &sc, const Surf<FAT> &s, float cx, float cy, V3 &vcol, V3 &nraw) {
    int prim = s.ref;
    if (FAT) {
        const float cz = 1.0f - cx - cy;
        // (attributes at floats 12..29 of the record, or -- plane records with index / mask first -- at 14..31)
        const V3 N0 = kRecMetaFirst ? mk(s.r3.z, s.r3.w, s.r4.x) : mk(s.r3.x, s.r3.y, s.r3.z), N1 = kRecMetaFirst ? mk(s.r4.y, s.r4.z, s.r4.w) : mk(s.r3.w, s.r4.x, s.r4.y);
        const V3 N2 = kRecMetaFirst ? mk(s.r5.x, s.r5.y, s.r5.z) : mk(s.r4.z, s.r4.w, s.r5.x);
        const V3 C0 = kRecMetaFirst ? mk(s.r5.w, s.r6.x, s.r6.y) : mk(s.r5.y, s.r5.z, s.r5.w), C1 = kRecMetaFirst ? mk(s.r6.z, s.r6.w, s.r7.x) : mk(s.r6.x, s.r6.y, s.r6.z);
        const V3 C2 = kRecMetaFirst ? mk(s.r7.y, s.r7.z, s.r7.w) : mk(s.r6.w, s.r7.x, s.r7.y);
        vcol = cx * C0 + cy * C1 + cz * C2;
        nraw = cx * N0 + cy * N1 + cz * N2;
        if (sc.tex.uv) prim = fat_prim(sc.tris, (uint32_t)s.ref);   // textures are addressed by the original index
    } else {
        vcol = interp_attr(sc.colors, prim, cx, cy);
        nraw = interp_attr(sc.normals, prim, cx, cy);
    }
    if (sc.tex.uv) vcol = texture_albedo(sc.tex, prim, cx, cy, vcol);   // wave-uniform test: scenes without textures skip it
}

// ACES + sRGB (N1): common.h:36-43,163-171
TRG_DEV float aces_film(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fminf(fmaxf((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f), 1.0f);
}
TRG_DEV float to_srgb(float v) {
    if (v < 0.0031308f) v *= 12.92f;
    else v = 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
    return v;
}

TRG_DEV uint32_t pcg_hash32(uint32_t v) {
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

}  // namespace trgdev
