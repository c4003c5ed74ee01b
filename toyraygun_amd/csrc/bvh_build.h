// bvh_build.h -- host-side acceleration-structure build for the HIP intersector.
//
// Replaces what the reference delegates to closed vendor code:
// MPSTriangleAccelerationStructure rebuild (src/engine/Metal/MetalRenderer.mm:272-279) /
// DXR BuildRaytracingAccelerationStructure (src/engine/D3D12/D3D12Renderer.cpp:285-391).
#pragma once
#include <cstdint>
#include <vector>

namespace trg {

struct F4 { float x, y, z, w; };

// Device layout (see DESIGN.md "Data layout in HBM"):
//  node i = 4 x float4 (64 B, one coalesced line half):
//    n[0] = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//    n[1] = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//    n[2] = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//    n[3] = (bits child0, bits child1, 0, 0)   child >= 0: inner node index
//                                               child <  0: leaf, ~child = (firstTri << 3) | (count-1), count <= 7;
//                                               (firstTri << 3) | 7 = a QUAD leaf (round 4): records firstTri and firstTri + 1 are the two
//                                               triangles of a parallelogram (leaf_count / kLeafQuad below)
//  every node has two children; the root is always an inner node (bvh_build.cpp synthesises one).
//  triangle record k (leaf order) = 3 x float4 (48 B):
//    t[0] = (v0.xyz, bits primitiveIndex)  t[1] = (e1.xyz, bits mask)  t[2] = (e2.xyz, 0)
//  wide node i (scenes kept in HBM) = 8 x float4 (128 B, one cache line), four children, SoA:
//    w[0..5] = lo.x[4], hi.x[4], lo.y[4], hi.y[4], lo.z[4], hi.z[4]   w[6] = (bits child0..3)   w[7] = padding
//    child encoding as above; an unused slot holds 0x80000000 and is never entered.
// QUAD leaves (round 4).  The reference's shapes are made of quads -- addPlane / addAreaLight / every cube face are two triangles that share an
// edge and span a parallelogram (Scene.cpp:24-92) -- and the shipped build's plane form tests a parallelogram as cheaply as a triangle.  The
// builder pairs such triangles (consecutive in the index buffer, sharing their first vertex and one more, fourth corner = the sum of the
// other two to a few ulps, equal material), treats a pair as ONE primitive and gives it a leaf of its own, marked by the count field 7.
// Its two records are stored X first, Y second such that in the coordinates  P = p0 + s * X.e1 + t * Y.e2  triangle X is the half s >= t with
// weights (u, v) = (s - t, t) and triangle Y the half t > s with (u, v) = (s, t - s).  The strict build (and anything that walks the tree
// without knowing) tests the two triangles one after the other: kLeafQuad = a count of 2.
constexpr uint32_t kLeafQuad = 7u;
inline uint32_t leaf_count(uint32_t code) { return (code & 7u) == kLeafQuad ? 2u : (code & 7u) + 1u; }
// BOX leaves (round 5; scenes staged in LDS only).  addCube's twelve triangles (Scene.cpp:24-58) are six quads that bound a parallelepiped.  With
// `want_boxes` the builder treats such a group as ONE primitive while it splits -- it gets a subtree of its own, six quad leaves whose records are
// consecutive -- and reports the subtree's root (BoxLeaf): whoever lays the nodes out for LDS may replace the reference to that root by the leaf
// code ~((first_rec << 3) | kLeafBox), a leaf of TWELVE records.  The strict build tests the twelve triangles; the shipped build does one slab
// test in the box's own frame and reads (triangle, barycentrics) off the face it enters by.  The tree every other consumer sees (the BVH2 node
// array, its 4-wide collapse) just holds the subtree.  A plain leaf therefore holds at most six triangles (count field 0..5).
constexpr uint32_t kLeafBox = 6u;
struct BoxLeaf {
    uint32_t node;          // index (BVH2 node array) of the root of the box's subtree
    uint32_t first_rec;     // its twelve records: six quads, X then Y each
    float center[3];        // the parallelepiped: local coordinate l_k = axis[k] . (P - center), inside <=> |l_k| <= 1 for k = 0, 1, 2
    float axis[3][3];
    uint8_t face_rec[6];    // face f = 2 k + (l_k > 0 ? 1 : 0): offset (0, 2, .. 10) of its quad's X record from first_rec
    uint8_t face_bits[6];   // how that quad's (s, t) -- P = p0 + s X.e1 + t Y.e2 -- follow the two OTHER local coordinates l_i, l_j (i < j) of a point on the face:
                            // bit 0: s follows l_j (and t l_i); bit 1: s = (1 - l) / 2 instead of (1 + l) / 2; bit 2: the same for t
    uint32_t mask;          // the material id of its twelve triangles
};

struct Bvh {
    std::vector<uint8_t> quad;   // per triangle record: 1 = record X of a quad leaf (the next record is its Y)
    uint32_t n_quads = 0;
    std::vector<F4> nodes;   // 4 per node (BVH2)
    std::vector<F4> nodes4;  // 8 per node (BVH4 collapse of the same tree), float boxes: what the quantiser and the tests see
    std::vector<uint32_t> nodes4q;  // 16 dwords per node: the quantised 64-byte form the HBM kernels traverse (q4node.h)
    std::vector<F4> tris;    // 3 per triangle
    uint32_t n_nodes = 0, n_leaves = 0, depth = 0, max_leaf = 0;
    uint32_t n_nodes4 = 0, depth4 = 0;
    std::vector<BoxLeaf> boxes;   // want_boxes: the parallelepipeds found (see kLeafBox), then -- node = ~0u -- the lone quads as boxes of no thickness (HBM flavour only)
    uint32_t n_boxes_real = 0;    // how many of them are parallelepipeds
    std::vector<uint32_t> nodes4q_box;   // the quantised 4-wide nodes once more, with every box a LEAF: child code ~(((number of records + index into `boxes`) << 3) | kLeafBox) -- what
    uint32_t n_nodes4_box = 0, depth4_box = 0;   // the shipped build traverses from HBM (empty: no boxes, or the whole scene is one)
    double sah_cost = 0.0;
    // ---- 8-wide compressed nodes (round-5 experiment, TRG_WIDE8 builds; q8node.h): the same BVH2 collapsed to up to eight children per node ----
    std::vector<uint32_t> nodes8;   // 20 dwords (80 bytes) per node, breadth first: the inner children of a node have consecutive indices
    std::vector<uint32_t> rec8;     // the leaf records in the order the 8-wide tree addresses them: rec8[i] = index into `tris` (record), or ~0u = a
                                    // never-hit padding record.  Every leaf owns TWO consecutive entries (one or two triangles, or the X and Y of a quad).
    std::vector<uint8_t> rec8_flags;  // per entry of rec8: bit 0 = this leaf's second entry holds a triangle to test too, bit 1 = the X of a quad
    uint32_t n_nodes8 = 0, depth8 = 0;
    bool wide8_ok = false;          // false: not built, or a leaf of more than two records (TRG_BVH_MAXLEAF > 2)
};

// the pairing rule by itself (the device builders use it too): are triangles k and k + 1 the two halves of a parallelogram?  x / y = which is X, which Y
bool quad_pair(const float *positions3, const uint32_t *indices, const uint32_t *masks, uint32_t ntris, uint32_t k, uint32_t &x, uint32_t &y);
bool quads_enabled();   // TRG_BVH_QUADS != 0 (default on), read at every call
// the primitives of a scene in index-buffer order: px[i] = the triangle (or the X of a quad), py[i] = the Y of a quad or ~0u; returns the number of quads
uint32_t pair_quads(const float *positions3, const uint32_t *indices, const uint32_t *masks, uint32_t ntris, std::vector<uint32_t> &px, std::vector<uint32_t> &py);

// positions3: nverts*3 floats; indices: ntris*3; masks: ntris (the reference's materialID buffer,
// MetalRenderer.mm:276).  Deterministic for a given input.
void build_bvh(const float *positions3, const uint32_t *indices, const uint32_t *masks, uint32_t ntris, Bvh &out, bool want_wide8 = false, bool want_boxes = false);
bool boxes_enabled();   // TRG_BVH_BOXES != 0 (default on), read at every call: the switch for A/B runs and tests

}  // namespace trg
