// trg_build.hip -- acceleration-structure build ON the GPU: LBVH (Morton codes + radix sort +
// Karras 2012 hierarchy + atomic bottom-up refit), emitted directly in the 4-wide node format the HBM
// traversal reads.  The device counterpart of what the reference delegates to MPSTriangleAccelerationStructure
// rebuild (src/engine/Metal/MetalRenderer.mm:272-279) / DXR BuildRaytracingAccelerationStructure
// (src/engine/D3D12/D3D12Renderer.cpp:285-391).  Trees are lower quality than the host SAH build (no surface-area
// heuristic) but take milliseconds for a million triangles; results of traversal are tree-independent
// (intersection contract: minimum t, ties to the lower primitive index), so images stay bit-identical.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "q4node.h"
#include "trg_build.h"

namespace trg {
namespace {

constexpr int kEmptyChild = (int)0x80000000;

struct Box3 { float lo[3], hi[3]; };

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// per triangle: padded box and the sort key (30-bit Morton code of the box centre << 32 | triangle index: unique)
__global__ void prim_kernel(const float *pos, const uint32_t *idx, uint32_t ntris, float3 slo, float3 sinv, float pad,
                            Box3 *boxes, unsigned long long *keys) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntris) return;
    Box3 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    for (int j = 0; j < 3; ++j) {
        const float *p = pos + (size_t)idx[k * 3 + j] * 3;
        for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], p[a]); b.hi[a] = fmaxf(b.hi[a], p[a]); }
    }
    const float cx = (0.5f * (b.lo[0] + b.hi[0]) - slo.x) * sinv.x;
    const float cy = (0.5f * (b.lo[1] + b.hi[1]) - slo.y) * sinv.y;
    const float cz = (0.5f * (b.lo[2] + b.hi[2]) - slo.z) * sinv.z;
    const uint32_t ix = (uint32_t)fminf(fmaxf(cx * 1024.0f, 0.0f), 1023.0f);
    const uint32_t iy = (uint32_t)fminf(fmaxf(cy * 1024.0f, 0.0f), 1023.0f);
    const uint32_t iz = (uint32_t)fminf(fmaxf(cz * 1024.0f, 0.0f), 1023.0f);
    const uint32_t morton = (expand_bits10(ix) << 2) | (expand_bits10(iy) << 1) | expand_bits10(iz);
    for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; }
    boxes[k] = b;
    keys[k] = ((unsigned long long)morton << 32) | k;
}

// triangle records in sorted (leaf) order: (v0, prim) (e1, mask) (e2, -), same arithmetic as the host builder
__global__ void record_kernel(const float *pos, const uint32_t *idx, const uint32_t *masks, const unsigned long long *keys,
                              uint32_t ntris, float4 *tris) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ntris) return;
    const uint32_t k = (uint32_t)(keys[r] & 0xFFFFFFFFull);
    const float *a = pos + (size_t)idx[k * 3 + 0] * 3, *b = pos + (size_t)idx[k * 3 + 1] * 3, *c = pos + (size_t)idx[k * 3 + 2] * 3;
    tris[(size_t)r * 3 + 0] = make_float4(a[0], a[1], a[2], __uint_as_float(k));
    tris[(size_t)r * 3 + 1] = make_float4(b[0] - a[0], b[1] - a[1], b[2] - a[2], __uint_as_float(masks[k]));
    tris[(size_t)r * 3 + 2] = make_float4(c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.0f);
}

__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));  // keys are unique, so the xor is never 0
}

// Karras 2012: internal node i of n-1; children encoded: >= 0 internal node, < 0 leaf ~r (r = sorted position)
__global__ void hierarchy_kernel(const unsigned long long *keys, int n, int *left, int *right, int *parent_int, int *parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int lc = (lo == gamma) ? ~gamma : gamma;
    const int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lc; right[i] = rc;
    if (lc >= 0) parent_int[lc] = i; else parent_leaf[~lc] = i;
    if (rc >= 0) parent_int[rc] = i; else parent_leaf[~rc] = i;
    if (i == 0) parent_int[0] = -1;
}

// bottom-up refit: the second thread to reach a node owns it
__global__ void refit_kernel(const Box3 *prim_boxes, const unsigned long long *keys, int n, const int *left, const int *right,
                             const int *parent_int, const int *parent_leaf, Box3 *node_boxes, int *arrive) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int node = parent_leaf[r];
    while (node >= 0) {
        __threadfence();  // release: our child boxes are written back before the arrival is published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // hipcc may drop the fence's own wait (cdna guide, G16 pitfall 12)
        if (atomicAdd(&arrive[node], 1) == 0) return;  // first arrival: the sibling subtree is not finished yet
        __threadfence();  // acquire: second arrival, make the sibling's writes visible before reading them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Box3 b;
        const int c[2] = { left[node], right[node] };
        for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
        for (int k = 0; k < 2; ++k) {
            const Box3 *cb = c[k] >= 0 ? &node_boxes[c[k]] : &prim_boxes[(uint32_t)(keys[~c[k]] & 0xFFFFFFFFull)];
            for (int a = 0; a < 3; ++a) {
                // L1-bypassing loads on top of the acquire fence
                b.lo[a] = fminf(b.lo[a], __builtin_nontemporal_load(&cb->lo[a]));
                b.hi[a] = fmaxf(b.hi[a], __builtin_nontemporal_load(&cb->hi[a]));
            }
        }
        node_boxes[node] = b;
        node = parent_int[node];
    }
}

// depth of every internal node (root = 0) by walking up; kept[i] = depth even = becomes a 4-wide node
// An internal node whose two children are single triangles never becomes a node of its own: the triangles sit next to
// each other in the sorted order, so whoever references it references a two-triangle leaf instead (the host builder's
// leaves hold up to two triangles as well).  The root is exempt.
__device__ __forceinline__ bool is_pair(const int *left, const int *right, int i) { return i > 0 && left[i] < 0 && right[i] < 0; }

// ---- greedy top-down collapse: level by level, every BVH2 node that becomes a 4-wide node
// starts from its two children and keeps opening the child with the largest surface area until four slots are full
// (pairs and leaves cannot be opened).  Inner slots become the wide nodes of the next level; their indices come from an
// exclusive scan, so the layout is deterministic, breadth-first, siblings adjacent, and every child index is larger than
// its parent's (which validate_wide_kernel checks: a tree that passes cannot send the traversal in a circle).
__device__ __forceinline__ float box_half_area(const Box3 &b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__global__ void collapse_plan_kernel(const int *frontier, int n_cur, const int *left, const int *right, const Box3 *node_boxes,
                                     int *slots4, uint32_t *cnt) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cur) return;
    const int i = frontier[j];
    int sl[4] = { left[i], right[i], kEmptyChild, kEmptyChild };
    int n = 2;
    while (n < 4) {
        int best = -1; float best_area = -1.0f;
        for (int k = 0; k < n; ++k)
            if (sl[k] >= 0 && !is_pair(left, right, sl[k])) {
                const float a = box_half_area(node_boxes[sl[k]]);
                if (a > best_area) { best_area = a; best = k; }
            }
        if (best < 0) break;
        const int c = sl[best];
        sl[best] = left[c];
        sl[n++] = right[c];
    }
    uint32_t m = 0;
    for (int k = 0; k < 4; ++k) {
        slots4[j * 4 + k] = sl[k];
        if (sl[k] >= 0 && !is_pair(left, right, sl[k])) ++m;
    }
    cnt[j] = m;
}
__global__ void collapse_emit_kernel(const int *frontier, int n_cur, const int *slots4, const uint32_t *off, uint32_t wide_base_cur,
                                     uint32_t wide_base_next, const int *left, const int *right, const Box3 *node_boxes,
                                     const Box3 *prim_boxes, const unsigned long long *keys, int *next_frontier, float4 *nodes4) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cur) return;
    float v[6][4]; int ref[4];
    uint32_t inner = 0;
    for (int k = 0; k < 4; ++k) {
        ref[k] = kEmptyChild;
        for (int a = 0; a < 6; ++a) v[a][k] = 0.0f;
        const int c = slots4[j * 4 + k];
        if (c == kEmptyChild) continue;
        const Box3 *b;
        if (c >= 0 && is_pair(left, right, c)) { b = &node_boxes[c]; ref[k] = ~(int)((((uint32_t)~left[c]) << 3) | 1u); }
        else if (c >= 0) { b = &node_boxes[c]; const uint32_t w = off[j] + inner++; ref[k] = (int)(wide_base_next + w); next_frontier[w] = c; }
        else { const uint32_t r = (uint32_t)~c; b = &prim_boxes[(uint32_t)(keys[r] & 0xFFFFFFFFull)]; ref[k] = ~(int)(r << 3); }
        for (int a = 0; a < 3; ++a) { v[a * 2][k] = b->lo[a]; v[a * 2 + 1][k] = b->hi[a]; }
    }
    float4 *o = nodes4 + (size_t)(wide_base_cur + (uint32_t)j) * 8;
    for (int a = 0; a < 6; ++a) o[a] = make_float4(v[a][0], v[a][1], v[a][2], v[a][3]);
    o[6] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), __int_as_float(ref[2]), __int_as_float(ref[3]));
    o[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
// structural check of a finished wide tree: inner children point forward and in range, leaves stay inside the records
__global__ void validate_wide_kernel(const float4 *nodes4, uint32_t n4, uint32_t ntris, int *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 c = nodes4[(size_t)i * 8 + 6];
    const int ref[4] = { __float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w) };
    for (int k = 0; k < 4; ++k) {
        if (ref[k] == kEmptyChild) continue;
        if (ref[k] >= 0) { if ((uint32_t)ref[k] <= i || (uint32_t)ref[k] >= n4) atomicExch(bad, 1); }
        else { const uint32_t code = (uint32_t)~ref[k]; if ((code >> 3) + (code & 7u) + 1u > ntris) atomicExch(bad, 1); }
    }
}

// float 4-wide nodes -> the 64-byte quantised nodes the kernels traverse (same encoder as the host build)
__global__ void quantize_nodes4_kernel(const float4 *nodes4, uint32_t n, uint4 *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float in[32];
    for (int r = 0; r < 8; ++r) { const float4 v = nodes4[(size_t)i * 8 + r]; in[r * 4] = v.x; in[r * 4 + 1] = v.y; in[r * 4 + 2] = v.z; in[r * 4 + 3] = v.w; }
    uint32_t q[16];
    trg::quantize_node4(in, q);
    for (int r = 0; r < 4; ++r) out[(size_t)i * 4 + r] = make_uint4(q[r * 4], q[r * 4 + 1], q[r * 4 + 2], q[r * 4 + 3]);
}

struct Tmp {
    void *p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};

#define BCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace

// Builds the 4-wide tree and the leaf-ordered triangle records for `ntris` >= 2 triangles.
// d_pos/d_idx/d_masks: device copies of the scene arrays.  d_nodes4 must hold (ntris - 1) * 8 float4 (upper
// bound on wide nodes), d_tris ntris * 3 float4.  Returns the number of wide nodes and the wide depth.
hipError_t gpu_build_lbvh(const float *d_pos, const uint32_t *d_idx, const uint32_t *d_masks, uint32_t ntris,
                          const float scene_lo[3], const float scene_hi[3], float pad, float4 *d_nodes4, float4 *d_tris,
                          uint32_t *n_nodes4, uint32_t *depth4, hipStream_t s) {
    const int n = (int)ntris, n_int = n - 1;
    const int T = 256;
    Tmp boxes, keys_a, keys_b, left, right, par_i, par_l, nboxes, arrive, kept, widx, maxd, sort_tmp, scan_tmp;
    BCHK(boxes.alloc(sizeof(Box3) * n)); BCHK(keys_a.alloc(8 * (size_t)n)); BCHK(keys_b.alloc(8 * (size_t)n));
    BCHK(left.alloc(4 * (size_t)n)); BCHK(right.alloc(4 * (size_t)n)); BCHK(par_i.alloc(4 * (size_t)n)); BCHK(par_l.alloc(4 * (size_t)n));
    BCHK(nboxes.alloc(sizeof(Box3) * n)); BCHK(arrive.alloc(4 * (size_t)n)); BCHK(kept.alloc(4 * (size_t)n)); BCHK(widx.alloc(4 * (size_t)n));
    BCHK(maxd.alloc(16));
    BCHK(hipMemsetAsync(arrive.p, 0, 4 * (size_t)n, s));
    BCHK(hipMemsetAsync(maxd.p, 0, 16, s));

    float3 slo = make_float3(scene_lo[0], scene_lo[1], scene_lo[2]);
    float3 sinv;
    sinv.x = scene_hi[0] > scene_lo[0] ? 1.0f / (scene_hi[0] - scene_lo[0]) : 0.0f;
    sinv.y = scene_hi[1] > scene_lo[1] ? 1.0f / (scene_hi[1] - scene_lo[1]) : 0.0f;
    sinv.z = scene_hi[2] > scene_lo[2] ? 1.0f / (scene_hi[2] - scene_lo[2]) : 0.0f;
    hipLaunchKernelGGL(prim_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, d_pos, d_idx, ntris, slo, sinv, pad, boxes.as<Box3>(),
                       keys_a.as<unsigned long long>());
    size_t tmp_bytes = 0;
    BCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys_a.as<unsigned long long>(), keys_b.as<unsigned long long>(), (size_t)n, 0, 62, s));
    BCHK(sort_tmp.alloc(tmp_bytes));
    BCHK(rocprim::radix_sort_keys(sort_tmp.p, tmp_bytes, keys_a.as<unsigned long long>(), keys_b.as<unsigned long long>(), (size_t)n, 0, 62, s));
    const unsigned long long *keys = keys_b.as<unsigned long long>();
    hipLaunchKernelGGL(record_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, d_pos, d_idx, d_masks, keys, ntris, d_tris);
    hipLaunchKernelGGL(hierarchy_kernel, dim3((n_int + T - 1) / T), dim3(T), 0, s, keys, n, left.as<int>(), right.as<int>(), par_i.as<int>(),
                       par_l.as<int>());
    hipLaunchKernelGGL(refit_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, boxes.as<Box3>(), keys, n, left.as<int>(), right.as<int>(),
                       par_i.as<int>(), par_l.as<int>(), nboxes.as<Box3>(), arrive.as<int>());
    uint32_t n4 = 0, d4 = 0;
    {
        // greedy level-by-level collapse (kept / widx double as the two frontier buffers)
        Tmp slots, cnt, off;
        BCHK(slots.alloc(16 * (size_t)n)); BCHK(cnt.alloc(4 * (size_t)n)); BCHK(off.alloc(4 * (size_t)n));
        tmp_bytes = 0;
        BCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, (size_t)n_int, rocprim::plus<uint32_t>(), s));
        BCHK(scan_tmp.alloc(tmp_bytes));
        int *front[2] = { kept.as<int>(), widx.as<int>() };
        BCHK(hipMemsetAsync(front[0], 0, 4, s));  // level 0: the root, BVH2 node 0
        uint32_t n_cur = 1, wide_base = 0;
        int cur = 0;
        while (n_cur > 0) {
            if (wide_base + n_cur > (uint32_t)n_int || d4 > 4096u) return hipErrorUnknown;  // cannot happen for a tree; never loop forever
            const dim3 g((n_cur + T - 1) / T);
            hipLaunchKernelGGL(collapse_plan_kernel, g, dim3(T), 0, s, front[cur], (int)n_cur, left.as<int>(), right.as<int>(), nboxes.as<Box3>(),
                               slots.as<int>(), cnt.as<uint32_t>());
            size_t tb = tmp_bytes;
            BCHK(rocprim::exclusive_scan(scan_tmp.p, tb, cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, (size_t)n_cur, rocprim::plus<uint32_t>(), s));
            uint32_t last_cnt = 0, last_off = 0;
            BCHK(hipMemcpyAsync(&last_cnt, cnt.as<uint32_t>() + (n_cur - 1), 4, hipMemcpyDeviceToHost, s));
            BCHK(hipMemcpyAsync(&last_off, off.as<uint32_t>() + (n_cur - 1), 4, hipMemcpyDeviceToHost, s));
            BCHK(hipStreamSynchronize(s));
            const uint32_t n_next = last_off + last_cnt;
            hipLaunchKernelGGL(collapse_emit_kernel, g, dim3(T), 0, s, front[cur], (int)n_cur, slots.as<int>(), off.as<uint32_t>(), wide_base,
                               wide_base + n_cur, left.as<int>(), right.as<int>(), nboxes.as<Box3>(), boxes.as<Box3>(), keys, front[cur ^ 1],
                               d_nodes4);
            BCHK(hipGetLastError());
            wide_base += n_cur;
            n_cur = n_next;
            cur ^= 1;
            ++d4;
        }
        n4 = wide_base;
    }
    // a malformed tree could send a traversal kernel in a circle: check the structure before anybody walks it
    BCHK(hipMemsetAsync(maxd.p, 0, 4, s));
    hipLaunchKernelGGL(validate_wide_kernel, dim3((n4 + T - 1) / T), dim3(T), 0, s, d_nodes4, n4, ntris, maxd.as<int>());
    int bad = 0;
    BCHK(hipMemcpyAsync(&bad, maxd.p, 4, hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    if (bad) return hipErrorUnknown;
    *n_nodes4 = n4;
    *depth4 = d4;
    return hipSuccess;
}

hipError_t gpu_quantize_nodes4(const float4 *d_nodes4, uint32_t n_nodes4, void *d_out, hipStream_t s) {
    if (n_nodes4 == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_nodes4_kernel, dim3((n_nodes4 + 127) / 128), dim3(128), 0, s, d_nodes4, n_nodes4, static_cast<uint4 *>(d_out));
    return hipGetLastError();
}

}  // namespace trg
