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

#include <stdio.h>
#include <stdlib.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "q4node.h"
#include "trg_build.h"
#include "trg_kernels.h"   // (TRG_REC_META_FIRST: the layout of the plane-form leaf records)

namespace trg {
namespace {

constexpr int kEmptyChild = (int)0x80000000;

struct Box3 { float lo[3], hi[3]; };

// spreads the low 21 bits of v to every third bit
__device__ __forceinline__ unsigned long long expand_bits21(uint32_t v) {
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// per triangle: padded box, the sort key (63-bit Morton code of the box centre) and the sort value (triangle index)
// (a primitive is a triangle, or -- px / py given -- the two triangles px[k], py[k] of a quad: bvh_build.h pair_quads; py[k] = ~0u: a lone triangle)
__global__ void prim_kernel(const float *pos, const uint32_t *idx, const uint32_t *px, const uint32_t *py, uint32_t nprims, float3 slo, float3 sinv, float pad,
                            Box3 *boxes, unsigned long long *codes, uint32_t *order) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nprims) return;
    Box3 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    const uint32_t t2[2] = { px ? px[k] : k, px ? py[k] : ~0u };
    for (int h = 0; h < 2; ++h) {
        if (t2[h] == ~0u) continue;
        for (int j = 0; j < 3; ++j) {
            const float *p = pos + (size_t)idx[t2[h] * 3 + j] * 3;
            for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], p[a]); b.hi[a] = fmaxf(b.hi[a], p[a]); }
        }
    }
    const float cx = (0.5f * (b.lo[0] + b.hi[0]) - slo.x) * sinv.x;
    const float cy = (0.5f * (b.lo[1] + b.hi[1]) - slo.y) * sinv.y;
    const float cz = (0.5f * (b.lo[2] + b.hi[2]) - slo.z) * sinv.z;
    const uint32_t ix = (uint32_t)fminf(fmaxf(cx * 2097152.0f, 0.0f), 2097151.0f);
    const uint32_t iy = (uint32_t)fminf(fmaxf(cy * 2097152.0f, 0.0f), 2097151.0f);
    const uint32_t iz = (uint32_t)fminf(fmaxf(cz * 2097152.0f, 0.0f), 2097151.0f);
    for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; }
    boxes[k] = b;
    codes[k] = (expand_bits21(ix) << 2) | (expand_bits21(iy) << 1) | expand_bits21(iz);
    order[k] = k;
}

// triangle records in sorted (leaf) order: (v0, prim) (e1, mask) (e2, -), same arithmetic as the host builder
// (quads: the primitive at sorted position r writes its X at record recbase[r] and its Y behind it, and flags the X record in quad_rec)
__global__ void record_kernel(const float *pos, const uint32_t *idx, const uint32_t *masks, const uint32_t *order, const uint32_t *px, const uint32_t *py,
                              const uint32_t *recbase, uint32_t nprims, float4 *tris, unsigned char *quad_rec) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nprims) return;
    const uint32_t prim = order[r];
    const uint32_t t2[2] = { px ? px[prim] : prim, px ? py[prim] : ~0u };
    const uint32_t rec0 = recbase ? recbase[r] : r;
    for (int h = 0; h < 2; ++h) {
        const uint32_t k = t2[h];
        if (k == ~0u) continue;
        const float *a = pos + (size_t)idx[k * 3 + 0] * 3, *b = pos + (size_t)idx[k * 3 + 1] * 3, *c = pos + (size_t)idx[k * 3 + 2] * 3;
        float4 *o = tris + (size_t)(rec0 + (uint32_t)h) * 3;
        o[0] = make_float4(a[0], a[1], a[2], __uint_as_float(k));
        o[1] = make_float4(b[0] - a[0], b[1] - a[1], b[2] - a[2], __uint_as_float(masks[k]));
        o[2] = make_float4(c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.0f);
    }
    if (quad_rec && t2[1] != ~0u) quad_rec[rec0] = 1;
}
// records a primitive occupies, by sorted position (the input of the exclusive scan that gives recbase)
__global__ void prim_size_kernel(const uint32_t *order, const uint32_t *py, uint32_t nprims, uint32_t *size) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nprims) size[r] = py[order[r]] != ~0u ? 2u : 1u;
}

// common-prefix length of sorted positions i and j; equal codes are told apart by the position itself
__device__ __forceinline__ int delta(const unsigned long long *codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = codes[i] ^ codes[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// Karras 2012: internal node i of n-1; children encoded: >= 0 internal node, < 0 leaf ~r (r = sorted position)
__global__ void hierarchy_kernel(const unsigned long long *codes, int n, int *left, int *right, int *parent_int, int *parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(codes, n, i, i + 1) - delta(codes, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(codes, n, i, i - d);
    int lmax = 2;
    while (delta(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(codes, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(codes, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(codes, n, i, i + (s + t) * d) > dnode) s += t;
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
__global__ void refit_kernel(const Box3 *prim_boxes, const uint32_t *order, int n, const int *left, const int *right,
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
            const Box3 *cb = c[k] >= 0 ? &node_boxes[c[k]] : &prim_boxes[order[~c[k]]];
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
// `root` is the BVH2 node the tree hangs from (0 for the Karras hierarchy, the last merge for PLOC); the two triangles must be
// neighbours in the record order (always true below a Karras node, checked for PLOC merges).
// (quads: recbase = the first record of every sorted position, one entry past the end included; a quad -- two records -- is a leaf of its own)
__device__ __forceinline__ bool is_quad_at(const uint32_t *recbase, uint32_t r) { return recbase && recbase[r + 1u] - recbase[r] == 2u; }
__device__ __forceinline__ bool is_pair(const int *left, const int *right, int i, int root, const uint32_t *recbase) {
#ifdef TRG_EXP_NO_PAIRS
    return false;
#endif
    if (!(i != root && left[i] < 0 && right[i] < 0 && (~right[i]) == (~left[i]) + 1)) return false;
    return !is_quad_at(recbase, (uint32_t)~left[i]) && !is_quad_at(recbase, (uint32_t)~right[i]);
}

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
                                     int *slots4, uint32_t *cnt, int root, const uint32_t *recbase) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cur) return;
    const int i = frontier[j];
    int sl[4] = { left[i], right[i], kEmptyChild, kEmptyChild };
    int n = 2;
    while (n < 4) {
        int best = -1; float best_area = -1.0f;
        for (int k = 0; k < n; ++k)
            if (sl[k] >= 0 && !is_pair(left, right, sl[k], root, recbase)) {
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
        if (sl[k] >= 0 && !is_pair(left, right, sl[k], root, recbase)) ++m;
    }
    cnt[j] = m;
}
__global__ void collapse_emit_kernel(const int *frontier, int n_cur, const int *slots4, const uint32_t *off, uint32_t wide_base_cur,
                                     uint32_t wide_base_next, const int *left, const int *right, const Box3 *node_boxes,
                                     const Box3 *prim_boxes, const uint32_t *order, int *next_frontier, float4 *nodes4, int root, const uint32_t *recbase) {
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
        if (c >= 0 && is_pair(left, right, c, root, recbase)) { b = &node_boxes[c]; const uint32_t r = (uint32_t)~left[c]; ref[k] = ~(int)(((recbase ? recbase[r] : r) << 3) | 1u); }
        else if (c >= 0) { b = &node_boxes[c]; const uint32_t w = off[j] + inner++; ref[k] = (int)(wide_base_next + w); next_frontier[w] = c; }
        else {   // one primitive: a triangle, or a quad (count field 7, bvh_build.h)
            const uint32_t r = (uint32_t)~c; b = &prim_boxes[order[r]];
            ref[k] = ~(int)(((recbase ? recbase[r] : r) << 3) | (is_quad_at(recbase, r) ? 7u : 0u));
        }
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
        else { const uint32_t code = (uint32_t)~ref[k]; if ((code >> 3) + ((code & 7u) == 7u ? 2u : (code & 7u) + 1u) > ntris) atomicExch(bad, 1); }
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

// ---------------------------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) on top of the Morton order: the clusters -- at first the
// triangles in Morton order -- each look for the neighbour within +-R positions whose union with them has the smallest surface
// area; mutual nearest neighbours merge into a BVH2 node; the survivors are compacted in order (one scan) and the round repeats
// until one cluster is left.  Unlike the Karras hierarchy, which splits where the Morton code says, every merge is chosen by
// area: the tree traces like a SAH tree (C4: DESIGN.md) and costs ~35 rounds of four small kernels for a million triangles.
// Node ids come from the scan, so the result is deterministic.
// ---------------------------------------------------------------------------------------------
#ifndef TRG_PLOC_RADIUS
#define TRG_PLOC_RADIUS 4
#endif
__device__ __forceinline__ float union_half_area(const Box3 &a, const Box3 &b) {
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}
__global__ void ploc_init_kernel(const Box3 *prim_boxes, const uint32_t *order, int n, int *cnode, Box3 *cbox) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    cnode[r] = ~r;   // leaf: record r
    cbox[r] = prim_boxes[order[r]];
}
__global__ void ploc_nn_kernel(const Box3 *cbox, int nc, int *nn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const Box3 b = cbox[i];
    int best = -1; float best_area = INFINITY;
    const int lo = max(0, i - TRG_PLOC_RADIUS), hi = min(nc - 1, i + TRG_PLOC_RADIUS);
    for (int j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const float a = union_half_area(b, cbox[j]);
        if (a < best_area) { best_area = a; best = j; }   // ties: the lower position
    }
    nn[i] = best;
}
// flags for the scan: low word = this cluster survives the round (everything but the upper partner of a merging pair), high word =
// this cluster is the lower partner of a merging pair (it becomes the new node)
__global__ void ploc_flag_kernel(const int *nn, int nc, unsigned long long *flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    flags[i] = (unsigned long long)((mutual && j < i) ? 0u : 1u) | ((unsigned long long)((mutual && i < j) ? 1u : 0u) << 32);
}
__global__ void ploc_merge_kernel(const int *nn, const unsigned long long *flags, const unsigned long long *offs, int nc, const int *cnode,
                                  const Box3 *cbox, int node_base, int *left, int *right, Box3 *node_boxes, int *cnode_out, Box3 *cbox_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const unsigned long long f = flags[i], o = offs[i];
    if ((f & 1ull) == 0ull) return;                    // absorbed by its partner
    const uint32_t pos = (uint32_t)(o & 0xFFFFFFFFull);
    if (f >> 32) {
        const int j = nn[i], id = node_base + (int)(o >> 32);
        Box3 u;
        for (int a = 0; a < 3; ++a) { u.lo[a] = fminf(cbox[i].lo[a], cbox[j].lo[a]); u.hi[a] = fmaxf(cbox[i].hi[a], cbox[j].hi[a]); }
        left[id] = cnode[i]; right[id] = cnode[j];     // i < j: Morton (= record) order is kept
        node_boxes[id] = u;
        cnode_out[pos] = id; cbox_out[pos] = u;
    } else {
        cnode_out[pos] = cnode[i]; cbox_out[pos] = cbox[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Binned-SAH top-down build, one tree level per round (the device twin of bvh_build.cpp's Builder::build: 16 centroid bins on each
// of the three axes, the split of least  area(L) * n(L) + area(R) * n(R)  over all 45 candidates, halving when the centroids
// coincide or below kSahDepthCap).  The triangles start in Morton order and every level is a STABLE partition of each active
// node's range (one scan over "goes left" flags), so a triangle's final position -- its record index -- is known as soon as its
// range has shrunk to one, and the result does not depend on timing: node ids and the next level's work list come from scans.
// Per-bin boxes are gathered with integer atomicMax on an order-preserving encoding of the floats (0 = empty, so one memset
// clears a level's bins).  Nodes of two triangles are finished by their parent (they become pair leaves in the collapse).
// The node boxes are NOT taken from the bins: sah_level_boxes_kernel recomputes them bottom-up, one launch per level.
// ---------------------------------------------------------------------------------------------
constexpr int kSahBins = 16;
constexpr int kSahBinWords = 7;                 // -lo xyz, hi xyz (encoded), count
constexpr int kSahNodeWords = 3 * kSahBins * kSahBinWords;
constexpr int kSahDepthCap = 24;

__device__ __forceinline__ uint32_t enc_ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float dec_ord(uint32_t v) { return __uint_as_float((v & 0x80000000u) ? (v & 0x7FFFFFFFu) : ~v); }

// centroid extent of active node j on axis a, and the bin of a triangle box in it (one function, used by the binning and by the partition)
__device__ __forceinline__ float sah_ext(const uint32_t *act_cbox, int j, int a) {
    return dec_ord(act_cbox[j * 6 + 3 + a]) + dec_ord(act_cbox[j * 6 + a]);   // hi - lo, lo kept negated
}
__device__ __forceinline__ float centroid(const Box3 &b, int a) { return 0.5f * (b.lo[a] + b.hi[a]); }
__device__ __forceinline__ int sah_bin(const uint32_t *act_cbox, int j, int a, const Box3 &b) {
    const float lo = -dec_ord(act_cbox[j * 6 + a]);
    const float scale = (float)kSahBins / sah_ext(act_cbox, j, a);
    return min(kSahBins - 1, max(0, (int)((centroid(b, a) - lo) * scale)));
}

// Both per-triangle passes below meet the same problem at the top of the tree: a million triangles send their atomics to the few
// words of one or two nodes.  Ranges are contiguous, so a workgroup's 256 triangles mostly belong to the node of its first triangle:
// that node is accumulated in LDS and flushed once per workgroup; triangles of other nodes (range boundaries, and most triangles once
// the nodes are smaller than a workgroup -- where nothing contends any more) go straight to memory.
constexpr int kSahBlock = 256;

// centroid bounds of every active node
__global__ __launch_bounds__(kSahBlock) void sah_cbox_kernel(const uint32_t *pid, const int *pnode, int n, const Box3 *boxes, uint32_t *act_cbox) {
    __shared__ uint32_t acc[6];
    __shared__ int j_first;
    const int p = blockIdx.x * kSahBlock + threadIdx.x;
    const int j = p < n ? pnode[p] : -1;
    if (threadIdx.x < 6) acc[threadIdx.x] = 0u;
    if (threadIdx.x == 0) j_first = j;
    __syncthreads();
    const int jb = j_first;
    if (j >= 0) {
        const Box3 b = boxes[pid[p]];
        for (int a = 0; a < 3; ++a) {
            const float c = centroid(b, a);
            if (j == jb) { atomicMax(&acc[a], enc_ord(-c)); atomicMax(&acc[3 + a], enc_ord(c)); }
            else { atomicMax(&act_cbox[j * 6 + a], enc_ord(-c)); atomicMax(&act_cbox[j * 6 + 3 + a], enc_ord(c)); }
        }
    }
    __syncthreads();
    if (jb >= 0 && threadIdx.x < 6) atomicMax(&act_cbox[jb * 6 + threadIdx.x], acc[threadIdx.x]);
}

__global__ __launch_bounds__(kSahBlock) void sah_bin_kernel(const uint32_t *pid, const int *pnode, int n, const Box3 *boxes, const uint32_t *act_cbox,
                                                          uint32_t *bins) {
    __shared__ uint32_t acc[kSahNodeWords];
    __shared__ int j_first;
    const int p = blockIdx.x * kSahBlock + threadIdx.x;
    const int j = p < n ? pnode[p] : -1;
    for (int i = threadIdx.x; i < kSahNodeWords; i += kSahBlock) acc[i] = 0u;
    if (threadIdx.x == 0) j_first = j;
    __syncthreads();
    const int jb = j_first;
    if (j >= 0) {
        const Box3 b = boxes[pid[p]];
        for (int a = 0; a < 3; ++a) {
            if (!(sah_ext(act_cbox, j, a) > 0.0f)) continue;
            const int o = (a * kSahBins + sah_bin(act_cbox, j, a, b)) * kSahBinWords;
            if (j == jb) {
                uint32_t *w = acc + o;
                for (int k = 0; k < 3; ++k) { atomicMax(&w[k], enc_ord(-b.lo[k])); atomicMax(&w[3 + k], enc_ord(b.hi[k])); }
                atomicAdd(&w[6], 1u);
            } else {
                uint32_t *w = bins + (size_t)j * kSahNodeWords + o;
                for (int k = 0; k < 3; ++k) { atomicMax(&w[k], enc_ord(-b.lo[k])); atomicMax(&w[3 + k], enc_ord(b.hi[k])); }
                atomicAdd(&w[6], 1u);
            }
        }
    }
    __syncthreads();
    if (jb < 0) return;
    for (int i = threadIdx.x; i < kSahNodeWords; i += kSahBlock) {
        const uint32_t v = acc[i];
        if (v == 0u) continue;                       // empty bin (an encoded float is never 0)
        uint32_t *w = bins + (size_t)jb * kSahNodeWords + i;
        if (i % kSahBinWords == 6) atomicAdd(w, v); else atomicMax(w, v);
    }
}

// one thread per active node: the split, and for the two children the flags the id scan needs (low word: an inner node, i.e. two or
// more triangles; high word: an active node of the next level, i.e. three or more)
__global__ void sah_split_kernel(int n_act, const int *act_count, const uint32_t *act_cbox, const uint32_t *bins, int level, int *sp_axis,
                                 int *sp_bin, int *sp_mid, unsigned long long *flags) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_act) return;
    const int count = act_count[j];
    float best_cost = INFINITY; int best_axis = -1, best_bin = -1, best_left = 0;
    if (level < kSahDepthCap) {
        for (int a = 0; a < 3; ++a) {
            if (!(sah_ext(act_cbox, j, a) > 0.0f)) continue;
            const uint32_t *w = bins + ((size_t)j * 3 + a) * (kSahBins * kSahBinWords);
            float right_area[kSahBins]; uint32_t right_cnt[kSahBins];
            float acc[6]; uint32_t cnt = 0;
            for (int k = 0; k < 6; ++k) acc[k] = -INFINITY;
#pragma unroll
            for (int b = kSahBins - 1; b > 0; --b) {
                const uint32_t c = w[b * kSahBinWords + 6];
                if (c) for (int k = 0; k < 6; ++k) acc[k] = fmaxf(acc[k], dec_ord(w[b * kSahBinWords + k]));
                cnt += c;
                const float dx = acc[3] + acc[0], dy = acc[4] + acc[1], dz = acc[5] + acc[2];   // hi - lo, lo kept negated
                right_area[b] = cnt ? dx * dy + dy * dz + dz * dx : 0.0f;
                right_cnt[b] = cnt;
            }
            for (int k = 0; k < 6; ++k) acc[k] = -INFINITY;
            cnt = 0;
#pragma unroll
            for (int b = 0; b < kSahBins - 1; ++b) {
                const uint32_t c = w[b * kSahBinWords + 6];
                if (c) for (int k = 0; k < 6; ++k) acc[k] = fmaxf(acc[k], dec_ord(w[b * kSahBinWords + k]));
                cnt += c;
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                const float dx = acc[3] + acc[0], dy = acc[4] + acc[1], dz = acc[5] + acc[2];
                const float cost = (dx * dy + dy * dz + dz * dx) * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; best_left = (int)cnt; }
            }
        }
    }
    const int mid = best_axis >= 0 ? best_left : count / 2;
    sp_axis[j] = best_axis; sp_bin[j] = best_bin; sp_mid[j] = mid;
    const int c2[2] = { mid, count - mid };
    for (int k = 0; k < 2; ++k)
        flags[2 * j + k] = (unsigned long long)(c2[k] >= 2 ? 1u : 0u) | ((unsigned long long)(c2[k] >= 3 ? 1u : 0u) << 32);
}

// one thread per active node: write the BVH2 node, finish children of one or two triangles, queue the others
__global__ void sah_emit_kernel(int n_act, const int *act_first, const int *act_count, const int *act_id, const int *sp_mid,
                                const unsigned long long *offs, int id_base_next, int *left, int *right,
                                int *next_first, int *next_count, int *next_id, int *child_act) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_act) return;
    const int id = act_id[j], first = act_first[j], count = act_count[j], mid = sp_mid[j];
    const int cf[2] = { first, first + mid }, cc[2] = { mid, count - mid };
    int ch[2];
    for (int k = 0; k < 2; ++k) {
        const unsigned long long o = offs[2 * j + k];
        int act = -1;
        if (cc[k] == 1) ch[k] = ~cf[k];
        else {
            const int cid = id_base_next + (int)(o & 0xFFFFFFFFull);
            ch[k] = cid;
            if (cc[k] == 2) {
                left[cid] = ~cf[k]; right[cid] = ~(cf[k] + 1);
            } else {
                act = (int)(o >> 32);
                next_first[act] = cf[k]; next_count[act] = cc[k]; next_id[act] = cid;
            }
        }
        child_act[2 * j + k] = act;
    }
    left[id] = ch[0]; right[id] = ch[1];
}

__global__ void sah_side_kernel(const uint32_t *pid, const int *pnode, int n, const Box3 *boxes, const uint32_t *act_cbox, const int *act_first,
                                const int *sp_axis, const int *sp_bin, const int *sp_mid, uint32_t *goes_left) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int j = pnode[p];
    uint32_t l = 0;
    if (j >= 0) {
        const int a = sp_axis[j];
        if (a >= 0) {
            l = sah_bin(act_cbox, j, a, boxes[pid[p]]) <= sp_bin[j] ? 1u : 0u;
        } else {
            l = (p - act_first[j]) < sp_mid[j] ? 1u : 0u;
        }
    }
    goes_left[p] = l;
}

// stable partition of every active range: lefts keep their order at the front, rights theirs behind them
__global__ void sah_scatter_kernel(const uint32_t *pid, const int *pnode, int n, const int *act_first, const int *sp_mid, const uint32_t *goes_left,
                                   const uint32_t *lefts_before, const int *child_act, uint32_t *pid_out, int *pnode_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int j = pnode[p];
    if (j < 0) { pid_out[p] = pid[p]; pnode_out[p] = -1; return; }
    const int f = act_first[j];
    const int lb = (int)(lefts_before[p] - lefts_before[f]);
    const int side = goes_left[p] ? 0 : 1;
    const int q = side == 0 ? f + lb : f + sp_mid[j] + (p - f - lb);
    pid_out[q] = pid[p];
    pnode_out[q] = child_act[2 * j + side];
}
__global__ void sah_init_kernel(int n, int *pnode, int *act_first, int *act_count, int *act_id) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) pnode[p] = 0;
    if (p == 0) { act_first[0] = 0; act_count[0] = n; act_id[0] = 0; }
}

// The host builder keeps two triangles in one leaf only when the SAH says so (bvh_build.cpp: leaf_cost <= split_cost with
// kTravCost = 1.2); otherwise they become two leaves of one triangle.  is_pair() reads "two adjacent leaves in record order" as a
// pair, so a node that should stay split gets its two children stored in descending order.
__global__ void sah_pair_kernel(int n_int, int *left, int *right, const Box3 *prim_boxes, const uint32_t *order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_int) return;
    const int l = left[i], r = right[i];
    if (!(l < 0 && r < 0 && (~r) == (~l) + 1)) return;
    const Box3 a = prim_boxes[order[~l]], b = prim_boxes[order[~r]];
    const float leaf_cost = union_half_area(a, b) * 2.0f;
    const float split_cost = 1.2f * union_half_area(a, b) + box_half_area(a) + box_half_area(b);
    if (!(leaf_cost <= split_cost)) { left[i] = r; right[i] = l; }
}

// Node boxes of the level-built tree, bottom-up one level per launch: every child of a node carries a larger id than any node of its
// parent's level, so when a level's launch runs its children's boxes are final -- no arrival counters, no fences.
__global__ void sah_level_boxes_kernel(int id_lo, int id_hi, const int *left, const int *right, const Box3 *prim_boxes, const uint32_t *order,
                                       Box3 *node_boxes) {
    const int i = id_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= id_hi) return;
    const int c[2] = { left[i], right[i] };
    Box3 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    for (int k = 0; k < 2; ++k) {
        const Box3 cb = c[k] >= 0 ? node_boxes[c[k]] : prim_boxes[order[~c[k]]];
        for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], cb.lo[a]); b.hi[a] = fmaxf(b.hi[a], cb.hi[a]); }
    }
    node_boxes[i] = b;
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
// Quads (round 4): d_px / d_py (device copies of bvh_build.h pair_quads' arrays, nprims entries) make the builders work on PRIMITIVES -- a lone
// triangle or the two triangles of a parallelogram, which become one leaf with the count field 7 and two records, X first; d_quad_rec (ntris bytes,
// zeroed by the caller) gets a 1 at every X record.  d_px == nullptr: every triangle is a primitive (nprims == ntris).
hipError_t gpu_build_lbvh(const float *d_pos, const uint32_t *d_idx, const uint32_t *d_masks, uint32_t ntris,
                          const float scene_lo[3], const float scene_hi[3], float pad, float4 *d_nodes4, float4 *d_tris,
                          uint32_t *n_nodes4, uint32_t *depth4, hipStream_t s, int mode,
                          const uint32_t *d_px, const uint32_t *d_py, uint32_t nprims, unsigned char *d_quad_rec) {
    if (!d_px) nprims = ntris;
    if (nprims < 2u || nprims > ntris) return hipErrorInvalidValue;
    const int n = (int)nprims, n_int = n - 1;
    const int T = 256;
    Tmp boxes, keys_a, keys_b, ord_a, ord_b, left, right, par_i, par_l, nboxes, arrive, kept, widx, maxd, sort_tmp, scan_tmp;
    BCHK(boxes.alloc(sizeof(Box3) * n)); BCHK(keys_a.alloc(8 * (size_t)n)); BCHK(keys_b.alloc(8 * (size_t)n));
    BCHK(ord_a.alloc(4 * (size_t)n)); BCHK(ord_b.alloc(4 * (size_t)n));
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
    hipLaunchKernelGGL(prim_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, d_pos, d_idx, d_px, d_py, nprims, slo, sinv, pad, boxes.as<Box3>(),
                       keys_a.as<unsigned long long>(), ord_a.as<uint32_t>());
    size_t tmp_bytes = 0;
    BCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a.as<unsigned long long>(), keys_b.as<unsigned long long>(), ord_a.as<uint32_t>(),
                                   ord_b.as<uint32_t>(), (size_t)n, 0, 63, s));
    BCHK(sort_tmp.alloc(tmp_bytes));
    BCHK(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, keys_a.as<unsigned long long>(), keys_b.as<unsigned long long>(), ord_a.as<uint32_t>(),
                                   ord_b.as<uint32_t>(), (size_t)n, 0, 63, s));
    const unsigned long long *codes = keys_b.as<unsigned long long>();   // sorted Morton codes
    const uint32_t *keys = ord_b.as<uint32_t>();                         // sorted position -> primitive index
    int root = 0;
    // quads: the first record of every sorted position (n + 1 entries: the last one is ntris)
    Tmp psize, recb, rec_tmp;
    const uint32_t *recbase = nullptr;
    auto make_recbase = [&](const uint32_t *order) -> hipError_t {
        if (!d_px) return hipSuccess;
        if (!psize.p) { BCHK(psize.alloc(4 * (size_t)(n + 1))); BCHK(recb.alloc(4 * (size_t)(n + 1))); }
        BCHK(hipMemsetAsync(psize.p, 0, 4 * (size_t)(n + 1), s));
        hipLaunchKernelGGL(prim_size_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, order, d_py, nprims, psize.as<uint32_t>());
        size_t rb = 0;
        BCHK(rocprim::exclusive_scan(nullptr, rb, psize.as<uint32_t>(), recb.as<uint32_t>(), 0u, (size_t)(n + 1), rocprim::plus<uint32_t>(), s));
        if (!rec_tmp.p) BCHK(rec_tmp.alloc(rb));
        BCHK(rocprim::exclusive_scan(rec_tmp.p, rb, psize.as<uint32_t>(), recb.as<uint32_t>(), 0u, (size_t)(n + 1), rocprim::plus<uint32_t>(), s));
        recbase = recb.as<uint32_t>();
        return hipSuccess;
    };
    if (mode == 0) {
        hipLaunchKernelGGL(hierarchy_kernel, dim3((n_int + T - 1) / T), dim3(T), 0, s, codes, n, left.as<int>(), right.as<int>(), par_i.as<int>(),
                           par_l.as<int>());
        hipLaunchKernelGGL(refit_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, boxes.as<Box3>(), keys, n, left.as<int>(), right.as<int>(),
                           par_i.as<int>(), par_l.as<int>(), nboxes.as<Box3>(), arrive.as<int>());
    } else if (mode == 1) {
        // binned SAH, level by level
        int level_base[kSahDepthCap + 48], n_levels = 0;   // level L holds the node ids [level_base[L], level_base[L + 1])
        level_base[n_levels++] = 0; level_base[n_levels++] = 1;
        if (n == 2) {
            const int l2[2] = { ~0, ~1 };
            BCHK(hipMemcpyAsync(left.p, &l2[0], 4, hipMemcpyHostToDevice, s)); BCHK(hipMemcpyAsync(right.p, &l2[1], 4, hipMemcpyHostToDevice, s));
            BCHK(hipStreamSynchronize(s));
        } else {
            const size_t a_max = (size_t)n / 3 + 1;    // an active node holds three triangles or more
            Tmp pn[2], af[2], ac[2], ai[2], acb, bins, spa, spb, spm, flags, offs, cact, gl, lb, sah_tmp;
            for (int k = 0; k < 2; ++k) {
                BCHK(pn[k].alloc(4 * (size_t)n)); BCHK(af[k].alloc(4 * a_max)); BCHK(ac[k].alloc(4 * a_max)); BCHK(ai[k].alloc(4 * a_max));
            }
            BCHK(acb.alloc(24 * a_max)); BCHK(bins.alloc(4 * (size_t)kSahNodeWords * a_max));
            BCHK(spa.alloc(4 * a_max)); BCHK(spb.alloc(4 * a_max)); BCHK(spm.alloc(4 * a_max));
            BCHK(flags.alloc(16 * a_max)); BCHK(offs.alloc(16 * a_max)); BCHK(cact.alloc(8 * a_max));
            BCHK(gl.alloc(4 * (size_t)n)); BCHK(lb.alloc(4 * (size_t)n));
            size_t b1 = 0, b2 = 0;
            BCHK(rocprim::exclusive_scan(nullptr, b1, flags.as<unsigned long long>(), offs.as<unsigned long long>(), 0ull, 2 * a_max,
                                         rocprim::plus<unsigned long long>(), s));
            BCHK(rocprim::exclusive_scan(nullptr, b2, gl.as<uint32_t>(), lb.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
            const size_t sah_bytes = b1 > b2 ? b1 : b2;
            BCHK(sah_tmp.alloc(sah_bytes));
            uint32_t *pid[2] = { ord_b.as<uint32_t>(), ord_a.as<uint32_t>() };   // the Morton order is the starting order
            const dim3 gp((n + T - 1) / T);
            hipLaunchKernelGGL(sah_init_kernel, gp, dim3(T), 0, s, n, pn[0].as<int>(), af[0].as<int>(), ac[0].as<int>(), ai[0].as<int>());
            int n_act = 1, id_base_next = 1, cur = 0, level = 0;
            while (n_act > 0) {
                if (level > kSahDepthCap + 40) { fprintf(stderr, "trg sah build: level cap\n"); return hipErrorUnknown; }   // below the cap every level halves the ranges
                const dim3 ga((n_act + T - 1) / T);
                BCHK(hipMemsetAsync(acb.p, 0, 24 * (size_t)n_act, s));
                BCHK(hipMemsetAsync(bins.p, 0, 4 * (size_t)kSahNodeWords * (size_t)n_act, s));
                hipLaunchKernelGGL(sah_cbox_kernel, gp, dim3(T), 0, s, pid[cur], pn[cur].as<int>(), n, boxes.as<Box3>(), acb.as<uint32_t>());
                hipLaunchKernelGGL(sah_bin_kernel, gp, dim3(T), 0, s, pid[cur], pn[cur].as<int>(), n, boxes.as<Box3>(), acb.as<uint32_t>(),
                                   bins.as<uint32_t>());
                hipLaunchKernelGGL(sah_split_kernel, ga, dim3(T), 0, s, n_act, ac[cur].as<int>(), acb.as<uint32_t>(), bins.as<uint32_t>(), level,
                                   spa.as<int>(), spb.as<int>(), spm.as<int>(), flags.as<unsigned long long>());
                size_t tb = sah_bytes;
                BCHK(rocprim::exclusive_scan(sah_tmp.p, tb, flags.as<unsigned long long>(), offs.as<unsigned long long>(), 0ull, (size_t)(2 * n_act),
                                             rocprim::plus<unsigned long long>(), s));
                unsigned long long last_f = 0, last_o = 0;
                BCHK(hipMemcpyAsync(&last_f, flags.as<unsigned long long>() + (2 * n_act - 1), 8, hipMemcpyDeviceToHost, s));
                BCHK(hipMemcpyAsync(&last_o, offs.as<unsigned long long>() + (2 * n_act - 1), 8, hipMemcpyDeviceToHost, s));
                hipLaunchKernelGGL(sah_emit_kernel, ga, dim3(T), 0, s, n_act, af[cur].as<int>(), ac[cur].as<int>(), ai[cur].as<int>(), spm.as<int>(),
                                   offs.as<unsigned long long>(), id_base_next, left.as<int>(), right.as<int>(),
                                   af[cur ^ 1].as<int>(), ac[cur ^ 1].as<int>(), ai[cur ^ 1].as<int>(), cact.as<int>());
                hipLaunchKernelGGL(sah_side_kernel, gp, dim3(T), 0, s, pid[cur], pn[cur].as<int>(), n, boxes.as<Box3>(), acb.as<uint32_t>(),
                                   af[cur].as<int>(), spa.as<int>(), spb.as<int>(), spm.as<int>(), gl.as<uint32_t>());
                tb = sah_bytes;
                BCHK(rocprim::exclusive_scan(sah_tmp.p, tb, gl.as<uint32_t>(), lb.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
                hipLaunchKernelGGL(sah_scatter_kernel, gp, dim3(T), 0, s, pid[cur], pn[cur].as<int>(), n, af[cur].as<int>(), spm.as<int>(),
                                   gl.as<uint32_t>(), lb.as<uint32_t>(), cact.as<int>(), pid[cur ^ 1], pn[cur ^ 1].as<int>());
                BCHK(hipStreamSynchronize(s));
                const unsigned long long tot = last_f + last_o;
                const int inner = (int)(tot & 0xFFFFFFFFull), active = (int)(tot >> 32);
                if (id_base_next + inner > n_int || (size_t)active > a_max) { fprintf(stderr, "trg sah build: level %d: %d + %d inner nodes of %d, %d active\n", level, id_base_next, inner, n_int, active); return hipErrorUnknown; }
                id_base_next += inner;
                level_base[n_levels++] = id_base_next;
                n_act = active;
                cur ^= 1;
                ++level;
            }
            if (id_base_next != n_int) { fprintf(stderr, "trg sah build: %d inner nodes, expected %d\n", id_base_next, n_int); return hipErrorUnknown; }
            keys = pid[cur];
        }
        hipLaunchKernelGGL(sah_pair_kernel, dim3((n_int + T - 1) / T), dim3(T), 0, s, n_int, left.as<int>(), right.as<int>(), boxes.as<Box3>(), keys);
        for (int L = n_levels - 2; L >= 0; --L) {
            const int cnt = level_base[L + 1] - level_base[L];
            if (cnt > 0)
                hipLaunchKernelGGL(sah_level_boxes_kernel, dim3((cnt + T - 1) / T), dim3(T), 0, s, level_base[L], level_base[L + 1], left.as<int>(),
                                   right.as<int>(), boxes.as<Box3>(), keys, nboxes.as<Box3>());
        }
    } else {
        Tmp cn[2], cb[2], nnb, flags, offs, ploc_tmp;
        for (int k = 0; k < 2; ++k) { BCHK(cn[k].alloc(4 * (size_t)n)); BCHK(cb[k].alloc(sizeof(Box3) * (size_t)n)); }
        BCHK(nnb.alloc(4 * (size_t)n)); BCHK(flags.alloc(8 * (size_t)n)); BCHK(offs.alloc(8 * (size_t)n));
        size_t ploc_bytes = 0;
        BCHK(rocprim::exclusive_scan(nullptr, ploc_bytes, flags.as<unsigned long long>(), offs.as<unsigned long long>(), 0ull, (size_t)n,
                                     rocprim::plus<unsigned long long>(), s));
        BCHK(ploc_tmp.alloc(ploc_bytes));
        hipLaunchKernelGGL(ploc_init_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, boxes.as<Box3>(), keys, n, cn[0].as<int>(), cb[0].as<Box3>());
        int nc = n, node_base = 0, cur = 0, rounds = 0;
        while (nc > 1) {
            if (++rounds > 4096) return hipErrorUnknown;   // every round merges at least the globally closest pair
            const dim3 g((nc + T - 1) / T);
            hipLaunchKernelGGL(ploc_nn_kernel, g, dim3(T), 0, s, cb[cur].as<Box3>(), nc, nnb.as<int>());
            hipLaunchKernelGGL(ploc_flag_kernel, g, dim3(T), 0, s, nnb.as<int>(), nc, flags.as<unsigned long long>());
            size_t tb = ploc_bytes;
            BCHK(rocprim::exclusive_scan(ploc_tmp.p, tb, flags.as<unsigned long long>(), offs.as<unsigned long long>(), 0ull, (size_t)nc,
                                         rocprim::plus<unsigned long long>(), s));
            unsigned long long last_f = 0, last_o = 0;
            BCHK(hipMemcpyAsync(&last_f, flags.as<unsigned long long>() + (nc - 1), 8, hipMemcpyDeviceToHost, s));
            BCHK(hipMemcpyAsync(&last_o, offs.as<unsigned long long>() + (nc - 1), 8, hipMemcpyDeviceToHost, s));
            hipLaunchKernelGGL(ploc_merge_kernel, g, dim3(T), 0, s, nnb.as<int>(), flags.as<unsigned long long>(), offs.as<unsigned long long>(), nc,
                               cn[cur].as<int>(), cb[cur].as<Box3>(), node_base, left.as<int>(), right.as<int>(), nboxes.as<Box3>(),
                               cn[cur ^ 1].as<int>(), cb[cur ^ 1].as<Box3>());
            BCHK(hipStreamSynchronize(s));
            const unsigned long long tot = last_f + last_o;
            const int survivors = (int)(tot & 0xFFFFFFFFull), merges = (int)(tot >> 32);
            if (merges <= 0 || survivors != nc - merges || node_base + merges > n_int) return hipErrorUnknown;
            node_base += merges;
            nc = survivors;
            cur ^= 1;
        }
        if (node_base != n_int) return hipErrorUnknown;
        root = n_int - 1;   // the last merge
    }
    // the triangle records, in leaf order (`keys` is final here: the SAH builder re-orders the primitives level by level)
    BCHK(make_recbase(keys));
    hipLaunchKernelGGL(record_kernel, dim3((n + T - 1) / T), dim3(T), 0, s, d_pos, d_idx, d_masks, keys, d_px, d_py, recbase, nprims, d_tris, d_quad_rec);
    uint32_t n4 = 0, d4 = 0;
    {
        // greedy level-by-level collapse (kept / widx double as the two frontier buffers)
        Tmp slots, cnt, off;
        BCHK(slots.alloc(16 * (size_t)n)); BCHK(cnt.alloc(4 * (size_t)n)); BCHK(off.alloc(4 * (size_t)n));
        tmp_bytes = 0;
        BCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, (size_t)n_int, rocprim::plus<uint32_t>(), s));
        BCHK(scan_tmp.alloc(tmp_bytes));
        int *front[2] = { kept.as<int>(), widx.as<int>() };
        BCHK(hipMemcpyAsync(front[0], &root, 4, hipMemcpyHostToDevice, s));  // level 0: the root of the BVH2
        BCHK(hipStreamSynchronize(s));
        uint32_t n_cur = 1, wide_base = 0;
        int cur = 0;
        while (n_cur > 0) {
            if (wide_base + n_cur > (uint32_t)n_int || d4 > 4096u) return hipErrorUnknown;  // cannot happen for a tree; never loop forever
            const dim3 g((n_cur + T - 1) / T);
            hipLaunchKernelGGL(collapse_plan_kernel, g, dim3(T), 0, s, front[cur], (int)n_cur, left.as<int>(), right.as<int>(), nboxes.as<Box3>(),
                               slots.as<int>(), cnt.as<uint32_t>(), root, recbase);
            size_t tb = tmp_bytes;
            BCHK(rocprim::exclusive_scan(scan_tmp.p, tb, cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, (size_t)n_cur, rocprim::plus<uint32_t>(), s));
            uint32_t last_cnt = 0, last_off = 0;
            BCHK(hipMemcpyAsync(&last_cnt, cnt.as<uint32_t>() + (n_cur - 1), 4, hipMemcpyDeviceToHost, s));
            BCHK(hipMemcpyAsync(&last_off, off.as<uint32_t>() + (n_cur - 1), 4, hipMemcpyDeviceToHost, s));
            BCHK(hipStreamSynchronize(s));
            const uint32_t n_next = last_off + last_cnt;
            hipLaunchKernelGGL(collapse_emit_kernel, g, dim3(T), 0, s, front[cur], (int)n_cur, slots.as<int>(), off.as<uint32_t>(), wide_base,
                               wide_base + n_cur, left.as<int>(), right.as<int>(), nboxes.as<Box3>(), boxes.as<Box3>(), keys, front[cur ^ 1],
                               d_nodes4, root, recbase);
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
    if (bad) { fprintf(stderr, "trg gpu build: wide tree failed validation\n"); return hipErrorUnknown; }
    *n_nodes4 = n4;
    *depth4 = d4;
    return hipSuccess;
}

// 48-byte geometry records in leaf order + the attribute arrays in original order -> the 128-byte leaf records the HBM kernels read
// (trg_device.h kRecV4): rows 0..2 the geometry record, floats 12..20 the triangle's normals, 21..29 its colours.  Two threads per
// record, 64 bytes each: coalesced stores, gathered 36-byte reads.
// the three planes of a triangle given as (v0, e1, e2): the arithmetic of trg_capi.cpp fill_plane_record, in double
__device__ void plane_rows(const float4 r0, const float4 r1, const float4 r2, float cx, float cy, float cz, float4 *o) {
    const double v0[3] = { (double)r0.x - cx, (double)r0.y - cy, (double)r0.z - cz }, e1[3] = { r1.x, r1.y, r1.z }, e2[3] = { r2.x, r2.y, r2.z };
    double n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    o[0] = make_float4(0.f, 0.f, 0.f, 1.f); o[1] = make_float4(0.f, 0.f, 0.f, -1.f); o[2] = make_float4(0.f, 0.f, 0.f, -1.f);   // never hit
    if (!(len > 0.0) || !isfinite(len)) return;
    for (int k = 0; k < 3; ++k) n[k] /= len;
    double a1[3] = { e2[1] * n[2] - e2[2] * n[1], e2[2] * n[0] - e2[0] * n[2], e2[0] * n[1] - e2[1] * n[0] };
    double a2[3] = { n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0] };
    const double s1 = e1[0] * a1[0] + e1[1] * a1[1] + e1[2] * a1[2], s2 = e2[0] * a2[0] + e2[1] * a2[1] + e2[2] * a2[2];
    if (s1 == 0.0 || s2 == 0.0) return;
    for (int k = 0; k < 3; ++k) { a1[k] /= s1; a2[k] /= s2; }
    const float4 p0 = make_float4((float)n[0], (float)n[1], (float)n[2], (float)(n[0] * v0[0] + n[1] * v0[1] + n[2] * v0[2]));
    const float4 p1 = make_float4((float)a1[0], (float)a1[1], (float)a1[2], (float)-(a1[0] * v0[0] + a1[1] * v0[1] + a1[2] * v0[2]));
    const float4 p2 = make_float4((float)a2[0], (float)a2[1], (float)a2[2], (float)-(a2[0] * v0[0] + a2[1] * v0[1] + a2[2] * v0[2]));
    const float f[12] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w };
    bool ok = true;
    for (int k = 0; k < 12; ++k) ok = ok && isfinite(f[k]);
    if (ok) { o[0] = p0; o[1] = p1; o[2] = p2; }
}

__global__ void fatten_records_kernel(const float4 *tris48, const float *nrm, const float *col, uint32_t n, float4 *out, bool planes, float cx, float cy, float cz,
                                      const unsigned char *quad_rec) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t >> 1, half = t & 1u;
    if (i >= n) return;
    const float4 r0 = tris48[(size_t)i * 3];
    const uint32_t prim = __float_as_uint(r0.w);
    const float *N = nrm + (size_t)prim * 9, *C = col + (size_t)prim * 9;
    float4 *o = out + (size_t)i * 8;
    if (half == 0u) {
        const float4 r1 = tris48[(size_t)i * 3 + 1];
        float4 r2 = tris48[(size_t)i * 3 + 2];
        // the X record of a quad carries the parallelogram's planes: X.e1 and Y.e2 (trg_capi.cpp fill_plane_record)
        if (planes && quad_rec && quad_rec[i] && i + 1u < n) r2 = tris48[(size_t)(i + 1u) * 3 + 2];
        if (planes) plane_rows(r0, r1, r2, cx, cy, cz, o);
        else { o[0] = r0; o[1] = r1; o[2] = r2; }
        // the plane form has no room for them in rows 0..2: the original index and the material id follow the planes (TRG_REC_META_FIRST)
        // or close the record
        if (planes && TRG_REC_META_FIRST) o[3] = make_float4(r0.w, r1.w, N[0], N[1]);
        else o[3] = make_float4(N[0], N[1], N[2], N[3]);
    } else if (planes && TRG_REC_META_FIRST) {
        o[4] = make_float4(N[2], N[3], N[4], N[5]);
        o[5] = make_float4(N[6], N[7], N[8], C[0]);
        o[6] = make_float4(C[1], C[2], C[3], C[4]);
        o[7] = make_float4(C[5], C[6], C[7], C[8]);
    } else {
        o[4] = make_float4(N[4], N[5], N[6], N[7]);
        o[5] = make_float4(N[8], C[0], C[1], C[2]);
        o[6] = make_float4(C[3], C[4], C[5], C[6]);
        o[7] = planes ? make_float4(C[7], C[8], r0.w, tris48[(size_t)i * 3 + 1].w) : make_float4(C[7], C[8], 0.0f, 0.0f);
    }
}

hipError_t gpu_fatten_records(const float4 *d_tris48, const float *d_normals, const float *d_colors, uint32_t ntris, void *d_out, bool planes, const float center[3], hipStream_t s,
                              const unsigned char *d_quad_rec) {
    if (ntris == 0) return hipSuccess;
    const uint64_t threads = 2ull * ntris;
    hipLaunchKernelGGL(fatten_records_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, d_tris48, d_normals, d_colors, ntris, static_cast<float4 *>(d_out), planes, center[0], center[1], center[2],
                       d_quad_rec);
    return hipGetLastError();
}

hipError_t gpu_quantize_nodes4(const float4 *d_nodes4, uint32_t n_nodes4, void *d_out, hipStream_t s) {
    if (n_nodes4 == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_nodes4_kernel, dim3((n_nodes4 + 127) / 128), dim3(128), 0, s, d_nodes4, n_nodes4, static_cast<uint4 *>(d_out));
    return hipGetLastError();
}

}  // namespace trg
