// bvh_build.cpp -- binned-SAH BVH2 builder, flattened into the "two child boxes per node" layout the
// gfx950 traversal kernel reads with four 16-byte loads (see bvh_build.h).
#include "bvh_build.h"
#include "q4node.h"
#include "q8node.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <limits>
#include <thread>

namespace trg {
namespace {

constexpr int kBins = 16;
uint32_t kMaxLeaf = 2;               // leaf encoding allows 6 (count field 6 = a BOX leaf, 7 = a QUAD leaf; tunable: TRG_BVH_MAXLEAF)
bool kQuads = true;                  // pair the two triangles of a parallelogram into one primitive with a leaf of its own (TRG_BVH_QUADS=0: off)
float kTravCost = 1.2f;               // SAH cost of one node visit relative to one triangle test (TRG_BVH_TRAVCOST)
constexpr uint32_t kSahDepthCap = 24; // below this depth switch to balanced median splits
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Box {
    float lo[3] = { kInf, kInf, kInf }, hi[3] = { -kInf, -kInf, -kInf };
    void grow(const float *p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim { Box b; float c[3]; uint32_t id; uint32_t id2; uint32_t box; };   // id2 != ~0u: a quad -- triangle id is its X, id2 its Y (bvh_build.h); box != ~0u: the twelve triangles from id on are a parallelepiped (kLeafBox)

struct BuildNode {
    Box box;
    int32_t child[2] = { -1, -1 };  // build-node indices, or -1
    uint32_t first = 0, count = 0;  // leaf range in prims[]
    bool leaf = false;
    uint32_t d = 0;                 // leaf: its depth
    uint32_t boxprim = ~0u;         // leaf inside the subtree of a box: the box's index in prims[]; `first` is then the quad's place (0..5) among the box's six
};

// A subtree handed to a worker thread: its triangle range, its depth, and the child slot of the top tree it hangs from.
struct Task { uint32_t first, count, depth; int32_t parent, slot; };

struct Builder {
    Prim *prims = nullptr;             // the builder sorts sub-ranges of ONE shared array in place (disjoint per subtree)
    std::vector<BuildNode> nodes;
    uint32_t depth = 0, leaves = 0, max_leaf = 0;
    uint32_t cutoff = 0;               // top tree only: ranges at most this long become tasks instead of being built
    std::vector<Task> *tasks = nullptr;

    int32_t make_leaf(const Box &box, uint32_t first, uint32_t count, uint32_t d) {
        BuildNode n; n.box = box; n.first = first; n.count = count; n.leaf = true; n.d = d;
        nodes.push_back(n);
        depth = std::max(depth, d); ++leaves; max_leaf = std::max(max_leaf, count);
        return (int32_t)nodes.size() - 1;
    }

    int32_t build(uint32_t first, uint32_t count, uint32_t d) {
        Box box, cbox;
        for (uint32_t i = 0; i < count; ++i) { box.grow(prims[first + i].b); cbox.grow(prims[first + i].c); }
        if (count <= 1) return make_leaf(box, first, count, d);
        // a quad (and a box) is a leaf of its own: a range that holds one is split until it is alone
        bool has_quad = false;
        for (uint32_t i = 0; i < count && !has_quad; ++i) has_quad = prims[first + i].id2 != ~0u || prims[first + i].box != ~0u;

        int axis = 0;
        float ext[3] = { cbox.hi[0] - cbox.lo[0], cbox.hi[1] - cbox.lo[1], cbox.hi[2] - cbox.lo[2] };
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;

        uint32_t mid = 0;
        bool have_split = false;
        if (d < kSahDepthCap && ext[axis] > 0.f) {
            // binned SAH over all three axes
            float best_cost = kInf; int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; ++a) {
                if (!(ext[a] > 0.f)) continue;
                Box bb[kBins]; uint32_t bc[kBins] = { 0 };
                const float scale = (float)kBins / ext[a];
                for (uint32_t i = 0; i < count; ++i) {
                    const Prim &p = prims[first + i];
                    int b = std::min(kBins - 1, std::max(0, (int)((p.c[a] - cbox.lo[a]) * scale)));
                    bb[b].grow(p.b); ++bc[b];
                }
                float right_area[kBins]; uint32_t right_cnt[kBins];
                Box acc; uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = cnt; }
                acc = Box(); cnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bb[b]); cnt += bc[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const float leaf_cost = box.half_area() * (float)count;
                // node traversal cost 1 box-pair test ~ 1.2 triangle tests
                const float split_cost = kTravCost * box.half_area() + best_cost;
                if (!has_quad && count <= kMaxLeaf && leaf_cost <= split_cost) return make_leaf(box, first, count, d);
                const float scale = (float)kBins / ext[best_axis];
                const float lo = cbox.lo[best_axis];
                auto it = std::partition(prims + first, prims + first + count, [&](const Prim &p) {
                    int b = std::min(kBins - 1, std::max(0, (int)((p.c[best_axis] - lo) * scale)));
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - (prims + first));
                have_split = mid > 0 && mid < count;
            }
        }
        if (!have_split) {
            if (!has_quad && count <= kMaxLeaf && (ext[axis] <= 0.f || d >= kSahDepthCap)) return make_leaf(box, first, count, d);
            // balanced median split (also the fallback for coincident centroids)
            mid = count / 2;
            std::nth_element(prims + first, prims + first + mid, prims + first + count,
                             [axis](const Prim &a, const Prim &b) {
                                 if (a.c[axis] != b.c[axis]) return a.c[axis] < b.c[axis];
                                 return a.id < b.id;
                             });
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        nodes[me].box = box;
        const int32_t l = child(first, mid, d + 1, me, 0);
        const int32_t r = child(first + mid, count - mid, d + 1, me, 1);
        nodes[me].child[0] = l; nodes[me].child[1] = r;
        return me;
    }

    // build the child now, or (top tree of a threaded build) leave it to a worker: -1 until the subtree is spliced in
    int32_t child(uint32_t first, uint32_t count, uint32_t d, int32_t parent, int32_t slot) {
        if (tasks && count <= cutoff) { tasks->push_back(Task{ first, count, d, parent, slot }); return -1; }
        return build(first, count, d);
    }
};

inline float bits_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

}  // namespace

// Quads: triangles k and k + 1 with the same first vertex, one more shared vertex, equal material, and the fourth corner where a
// parallelogram has it.  Pattern 1 (cube faces, Scene.cpp:37-55): (a, b, c) (a, c, d) -- X = k, Y = k + 1.  Pattern 2 (planes and the
// light, Scene.cpp:60-92): (a, b, c) (a, d, b) -- X = k + 1, Y = k.  In both, X = (a, p1, diag), Y = (a, diag', p3) with diag = p1 + p3 - a.
bool quad_pair(const float *pos, const uint32_t *idx, const uint32_t *masks, uint32_t ntris, uint32_t k, uint32_t &x, uint32_t &y) {
    auto vtx = [&](uint32_t t, int j) { return &pos[(size_t)idx[t * 3 + j] * 3]; };
    auto same = [](const float *p, const float *q) { return p[0] == q[0] && p[1] == q[1] && p[2] == q[2]; };
    if (k + 1 >= ntris || masks[k] != masks[k + 1] || !same(vtx(k, 0), vtx(k + 1, 0))) return false;
    const float *a = vtx(k, 0), *p1, *dg, *p3;
    if (same(vtx(k, 2), vtx(k + 1, 1))) { x = k; y = k + 1; p1 = vtx(k, 1); dg = vtx(k, 2); p3 = vtx(k + 1, 2); }
    else if (same(vtx(k, 1), vtx(k + 1, 2))) { x = k + 1; y = k; p1 = vtx(k + 1, 1); dg = vtx(k, 1); p3 = vtx(k, 2); }
    else return false;
    float big = 0.f;
    for (int c = 0; c < 3; ++c) big = std::max({ big, std::fabs(a[c]), std::fabs(p1[c]), std::fabs(p3[c]), std::fabs(dg[c]) });
    for (int c = 0; c < 3; ++c)
        if (!(std::fabs((p1[c] - a[c]) + (p3[c] - a[c]) - (dg[c] - a[c])) <= 4e-6f * big)) return false;   // (also false for NaN)
    const float e1[3] = { p1[0] - a[0], p1[1] - a[1], p1[2] - a[2] }, e2[3] = { p3[0] - a[0], p3[1] - a[1], p3[2] - a[2] };
    const float n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    const float area2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    return area2 > 0.f && std::isfinite(area2);
}

bool quads_enabled() { const char *e = getenv("TRG_BVH_QUADS"); return !e || atoi(e) != 0; }

uint32_t pair_quads(const float *pos, const uint32_t *idx, const uint32_t *masks, uint32_t ntris, std::vector<uint32_t> &px, std::vector<uint32_t> &py) {
    px.clear(); py.clear();
    px.reserve(ntris); py.reserve(ntris);
    const bool on = quads_enabled();
    uint32_t quads = 0;
    for (uint32_t k = 0; k < ntris; ++k) {
        uint32_t x = 0, y = 0;
        if (on && quad_pair(pos, idx, masks, ntris, k, x, y)) { px.push_back(x); py.push_back(y); ++quads; ++k; }
        else { px.push_back(k); py.push_back(~0u); }
    }
    return quads;
}

bool boxes_enabled() { const char *e = getenv("TRG_BVH_BOXES"); return !e || atoi(e) != 0; }

namespace {
// Are triangles k .. k + 11 six quads that bound a parallelepiped (addCube, Scene.cpp:24-58)?  Geometry in double from the fp32 corners:
// centre C = mean of the six face centres, half axes h_k = half the difference of two opposite face centres, A = [h_0 h_1 h_2]^-1, and
// every corner of every face must sit at local coordinates (+-1, +-1, +-1) to 1e-5 with the face's own sign on its axis.
struct BoxGeom { double c[3], a[3][3]; uint32_t face_quad[6]; };   // face f = 2 k + (l_k > 0): which of the six quads (in index order) it is
bool box_group(const float *pos, const uint32_t *idx, const uint32_t *masks, uint32_t ntris, uint32_t k, BoxGeom &g) {
    if (k + 12u > ntris) return false;
    auto vtx = [&](uint32_t t, int j) { return &pos[(size_t)idx[t * 3 + j] * 3]; };
    double corner[6][4][3], fc[6][3], C[3] = { 0, 0, 0 };
    for (uint32_t q = 0; q < 6; ++q) {
        uint32_t x = 0, y = 0;
        if (masks[k + 2 * q] != masks[k] || !quad_pair(pos, idx, masks, ntris, k + 2 * q, x, y)) return false;
        const float *p[4] = { vtx(x, 0), vtx(x, 1), vtx(x, 2), vtx(y, 2) };   // a, p1, the diagonal corner, p3 (quad_pair)
        for (int a = 0; a < 3; ++a) {
            fc[q][a] = 0;
            for (int j = 0; j < 4; ++j) { corner[q][j][a] = p[j][a]; fc[q][a] += 0.25 * p[j][a]; }
            C[a] += fc[q][a] / 6.0;
        }
    }
    // opposite faces: fc_q + fc_q' = 2 C
    int opp[6] = { -1, -1, -1, -1, -1, -1 };
    double scale = 0;
    for (int q = 0; q < 6; ++q) for (int a = 0; a < 3; ++a) scale = std::max(scale, std::fabs(fc[q][a] - C[a]));
    if (!(scale > 0) || !std::isfinite(scale)) return false;
    for (int q = 0; q < 6; ++q) {
        double best = 1e300; int bq = -1;
        for (int r = 0; r < 6; ++r) {
            if (r == q) continue;
            double e = 0;
            for (int a = 0; a < 3; ++a) e = std::max(e, std::fabs(fc[q][a] + fc[r][a] - 2 * C[a]));
            if (e < best) { best = e; bq = r; }
        }
        if (bq < 0 || best > 1e-5 * scale) return false;
        opp[q] = bq;
    }
    double h[3][3];
    int axis_of[6], nax = 0;
    for (int q = 0; q < 6; ++q) axis_of[q] = -1;
    for (int q = 0; q < 6; ++q) {
        if (opp[opp[q]] != q) return false;
        if (axis_of[q] >= 0) continue;
        if (nax == 3) return false;
        for (int a = 0; a < 3; ++a) h[nax][a] = 0.5 * (fc[q][a] - fc[opp[q]][a]);
        axis_of[q] = axis_of[opp[q]] = nax;
        g.face_quad[2 * nax + 1] = (uint32_t)q; g.face_quad[2 * nax] = (uint32_t)opp[q];   // + side: the face at C + h_k
        ++nax;
    }
    if (nax != 3) return false;
    // A = H^-1 (rows a_k: a_k . h_j = delta_kj), H's columns the half axes
    auto cross = [](const double *u, const double *v, double *o) { o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0]; };
    double c12[3], c20[3], c01[3];
    cross(h[1], h[2], c12); cross(h[2], h[0], c20); cross(h[0], h[1], c01);
    const double det = h[0][0] * c12[0] + h[0][1] * c12[1] + h[0][2] * c12[2];
    if (!(std::fabs(det) > 1e-12 * scale * scale * scale) || !std::isfinite(det)) return false;
    for (int a = 0; a < 3; ++a) { g.a[0][a] = c12[a] / det; g.a[1][a] = c20[a] / det; g.a[2][a] = c01[a] / det; g.c[a] = C[a]; }
    for (int q = 0; q < 6; ++q) {
        const int kx = axis_of[q];
        const double side = g.face_quad[2 * kx + 1] == (uint32_t)q ? 1.0 : -1.0;
        double sum[3] = { 0, 0, 0 };
        for (int j = 0; j < 4; ++j) {
            for (int ax = 0; ax < 3; ++ax) {
                double l = 0;
                for (int a = 0; a < 3; ++a) l += g.a[ax][a] * (corner[q][j][a] - C[a]);
                if (!(std::fabs(std::fabs(l) - 1.0) <= 1e-5)) return false;       // (also false for NaN)
                if (ax == kx && l * side < 0) return false;
                sum[ax] += l;
            }
        }
        for (int ax = 0; ax < 3; ++ax)
            if (ax != kx && std::fabs(sum[ax]) > 1e-3) return false;             // the four corners are the four sign pairs of the other two axes
    }
    return true;
}
}  // namespace

void build_bvh(const float *pos, const uint32_t *idx, const uint32_t *masks, uint32_t ntris, Bvh &out, bool want_wide8, bool want_boxes) {
    if (const char *e = getenv("TRG_BVH_MAXLEAF")) kMaxLeaf = (uint32_t)std::min(6, std::max(1, atoi(e)));
    if (const char *e = getenv("TRG_BVH_TRAVCOST")) kTravCost = (float)atof(e);
    kQuads = quads_enabled();   // (read per build: the tests switch it)
    Builder B;
    std::vector<Prim> prim_store(ntris);
    B.prims = prim_store.data();
    Box scene;
    auto vtx = [&](uint32_t k, int j) { return &pos[(size_t)idx[k * 3 + j] * 3]; };
    auto quad_of = [&](uint32_t k, uint32_t &x, uint32_t &y) { return kQuads && quad_pair(pos, idx, masks, ntris, k, x, y); };
    uint32_t n_prims = 0;
    std::vector<BoxGeom> box_geom;
    const bool boxes = want_boxes && kQuads && boxes_enabled();
    for (uint32_t k = 0; k < ntris; ++k) {
        Prim &p = B.prims[n_prims++];
        p = Prim();
        p.id = k; p.id2 = ~0u; p.box = ~0u;
        BoxGeom bg;
        if (boxes && box_group(pos, idx, masks, ntris, k, bg)) {   // twelve triangles, one primitive
            p.box = (uint32_t)box_geom.size();
            box_geom.push_back(bg);
            for (uint32_t t = k; t < k + 12u; ++t)
                for (int j = 0; j < 3; ++j) p.b.grow(vtx(t, j));
            for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]);
            scene.grow(p.b);
            k += 11u;
            continue;
        }
        uint32_t x = 0, y = 0;
        const bool q = quad_of(k, x, y);
        if (q) { p.id = x; p.id2 = y; }
        for (uint32_t t = k; t <= k + (q ? 1u : 0u); ++t)
            for (int j = 0; j < 3; ++j) p.b.grow(vtx(t, j));
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]);
        scene.grow(p.b);
        if (q) ++k;
    }
    B.nodes.reserve((size_t)n_prims * 2 + 2);
    // Large scenes: the top of the tree is built here, subtrees of at most `cutoff` triangles by worker threads (each sorts
    // its own range of the shared array and fills its own node vector), then spliced in.  Every split decision depends on
    // the triangles of its range only, so the tree -- and, after the depth-first flatten below, the device layout -- is the
    // same as the single-threaded one, whatever the thread count or timing.
    unsigned n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("TRG_BVH_THREADS")) n_threads = (unsigned)std::min(64, std::max(1, atoi(e)));
    std::vector<Task> tasks;
    if (n_prims >= 65536u && n_threads > 1) {
        B.cutoff = std::max(4096u, n_prims / (n_threads * 8u));
        B.tasks = &tasks;
    }
    int32_t root = n_prims ? B.build(0, n_prims, 0) : -1;
    if (!tasks.empty()) {
        B.tasks = nullptr;
        std::vector<Builder> sub(tasks.size());
        std::vector<int32_t> sub_root(tasks.size(), -1);
        std::atomic<size_t> next{ 0 };
        auto work = [&]() {
            for (size_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1)) {
                sub[t].prims = B.prims;
                sub[t].nodes.reserve((size_t)tasks[t].count * 2 + 2);
                sub_root[t] = sub[t].build(tasks[t].first, tasks[t].count, tasks[t].depth);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < n_threads; ++k) pool.emplace_back(work);
        work();
        for (std::thread &th : pool) th.join();
        for (size_t t = 0; t < tasks.size(); ++t) {   // splice, in task order
            const int32_t off = (int32_t)B.nodes.size();
            for (BuildNode n : sub[t].nodes) {
                for (int k = 0; k < 2; ++k)
                    if (n.child[k] >= 0) n.child[k] += off;
                B.nodes.push_back(n);
            }
            B.nodes[tasks[t].parent].child[tasks[t].slot] = off + sub_root[t];
            B.depth = std::max(B.depth, sub[t].depth); B.leaves += sub[t].leaves; B.max_leaf = std::max(B.max_leaf, sub[t].max_leaf);
        }
    }

    // ---- boxes: the leaf of a box primitive becomes the root of a subtree over its six quads (built by the same rule).  The quads keep
    //      the order that build leaves them in: box_quads[b][j] = (X, Y) of the j-th, records first + 2 j and first + 2 j + 1 ----
    struct QuadIds { uint32_t x, y, q; };
    std::vector<std::array<QuadIds, 6>> box_quads(box_geom.size());
    std::vector<int32_t> box_root_node(box_geom.size(), -1);   // build-node index of the subtree's root
    if (!box_geom.empty()) {
        const size_t n_before = B.nodes.size();
        for (size_t n = 0; n < n_before; ++n) {
            if (!B.nodes[n].leaf || B.nodes[n].count != 1u || B.prims[B.nodes[n].first].box == ~0u) continue;
            const uint32_t bp = B.nodes[n].first, b = B.prims[bp].box, k0 = B.prims[bp].id;
            Prim lp[6];
            for (uint32_t q = 0; q < 6; ++q) {
                Prim &p = lp[q];
                p = Prim();
                uint32_t x = 0, y = 0;
                (void)quad_pair(pos, idx, masks, ntris, k0 + 2 * q, x, y);   // (box_group has checked it)
                p.id = x; p.id2 = y; p.box = ~0u;
                for (uint32_t t = k0 + 2 * q; t <= k0 + 2 * q + 1; ++t)
                    for (int j = 0; j < 3; ++j) p.b.grow(vtx(t, j));
                for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]);
            }
            Builder sub;
            sub.prims = lp;
            const int32_t sr = sub.build(0, 6, B.nodes[n].d);
            for (uint32_t j = 0; j < 6; ++j) box_quads[b][j] = QuadIds{ lp[j].id, lp[j].id2, (std::min(lp[j].id, lp[j].id2) - k0) / 2u };
            const int32_t off = (int32_t)B.nodes.size();
            for (BuildNode sn : sub.nodes) {
                for (int c = 0; c < 2; ++c)
                    if (sn.child[c] >= 0) sn.child[c] += off;
                if (sn.leaf) sn.boxprim = bp;      // (first = the quad's place among the six)
                B.nodes.push_back(sn);
            }
            B.nodes[n] = B.nodes[(size_t)(off + sr)];   // the box's leaf becomes the subtree's root: whoever pointed at it still does
            box_root_node[b] = (int32_t)n;
            B.depth = std::max(B.depth, sub.depth); B.leaves += sub.leaves - 1u; B.max_leaf = std::max(B.max_leaf, sub.max_leaf);
        }
    }

    // Conservative padding: the slab test and the triangle test round differently; every box is
    // widened by far more than that so a triangle the Moeller-Trumbore test accepts is always reached.
    float diag = 0.f;
    if (ntris) diag = std::max({ scene.hi[0] - scene.lo[0], scene.hi[1] - scene.lo[1], scene.hi[2] - scene.lo[2] });
    // ... and by a few ulps of the largest coordinate: the slab test evaluates plane * (1/d) - origin * (1/d), whose
    // rounding error grows with the distance from the origin, not with the size of the scene (a 2 cm object 10 m away)
    float maxabs = 0.f;
    if (ntris) for (int a = 0; a < 3; ++a) maxabs = std::max({ maxabs, std::fabs(scene.lo[a]), std::fabs(scene.hi[a]) });
    const float pad = std::max(2e-5f * std::max(diag, 1e-3f), 4e-6f * maxabs);

    // ---- flatten: inner build nodes become device nodes, in depth-first order ----
    std::vector<int32_t> dev_index(B.nodes.size(), -1);
    uint32_t n_inner = 0;
    // if the root is a leaf (or the scene is empty) synthesise an inner root
    const bool synth_root = (root < 0) || B.nodes[root].leaf;
    {
        std::vector<int32_t> st;
        if (!synth_root) st.push_back(root);
        while (!st.empty()) {
            int32_t n = st.back(); st.pop_back();
            dev_index[n] = (int32_t)n_inner++;
            // push right first so that the left child is laid out right after its parent
            for (int c = 1; c >= 0; --c) { int32_t ch = B.nodes[n].child[c]; if (ch >= 0 && !B.nodes[ch].leaf) st.push_back(ch); }
        }
    }
    out.n_nodes = synth_root ? 1 : n_inner;
    out.nodes.assign((size_t)out.n_nodes * 4, F4{ 0, 0, 0, 0 });
    out.tris.resize((size_t)ntris * 3);
    out.n_leaves = B.leaves; out.depth = B.depth + (synth_root ? 1 : 0); out.max_leaf = B.max_leaf;

    // triangle records in leaf (prims[]) order; a quad contributes its X record, then its Y record
    std::vector<uint32_t> rec_first((size_t)n_prims + 1, 0u);   // first record of prim i
    for (uint32_t i = 0; i < n_prims; ++i) rec_first[i + 1] = rec_first[i] + (B.prims[i].box != ~0u ? 12u : B.prims[i].id2 != ~0u ? 2u : 1u);
    out.quad.assign(ntris, 0);
    out.n_quads = 0;
    auto put_rec = [&](uint32_t r, uint32_t k) {
        const float *a = &pos[(size_t)idx[k * 3 + 0] * 3], *b = &pos[(size_t)idx[k * 3 + 1] * 3], *c = &pos[(size_t)idx[k * 3 + 2] * 3];
        out.tris[(size_t)r * 3 + 0] = F4{ a[0], a[1], a[2], bits_f(k) };
        out.tris[(size_t)r * 3 + 1] = F4{ b[0] - a[0], b[1] - a[1], b[2] - a[2], bits_f(masks[k]) };
        out.tris[(size_t)r * 3 + 2] = F4{ c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.f };
    };
    for (uint32_t i = 0; i < n_prims; ++i) {
        if (B.prims[i].box != ~0u) {   // six quads, in the order their subtree left them
            for (uint32_t j = 0; j < 6; ++j) {
                const QuadIds &qi = box_quads[B.prims[i].box][j];
                put_rec(rec_first[i] + 2 * j, qi.x); put_rec(rec_first[i] + 2 * j + 1u, qi.y);
                out.quad[rec_first[i] + 2 * j] = 1; ++out.n_quads;
            }
            continue;
        }
        put_rec(rec_first[i], B.prims[i].id);
        if (B.prims[i].id2 != ~0u) { put_rec(rec_first[i] + 1u, B.prims[i].id2); out.quad[rec_first[i]] = 1; ++out.n_quads; }
    }

    auto padded = [&](Box b) { for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; } return b; };
    // leaf over the prims [first, first + count): a quad leaf (one quad prim) or 1..7 single triangles
    auto leaf_ref = [&](uint32_t first, uint32_t count) {
        const uint32_t r0 = rec_first[first], nrec = rec_first[first + count] - r0;
        const bool q = count == 1u && B.prims[first].id2 != ~0u;
        return ~(int32_t)((r0 << 3) | (q ? kLeafQuad : nrec - 1u));
    };
    // ... and the leaf code of a leaf build node: a quad of a box's subtree addresses its pair of the box's twelve records
    auto leaf_of = [&](const BuildNode &n) {
        if (n.boxprim != ~0u) return ~(int32_t)(((rec_first[n.boxprim] + 2u * n.first) << 3) | kLeafQuad);
        return leaf_ref(n.first, n.count);
    };
    auto put = [&](uint32_t di, const Box &b0, int32_t r0, const Box &b1, int32_t r1) {
        F4 *n = &out.nodes[(size_t)di * 4];
        n[0] = F4{ b0.lo[0], b0.hi[0], b0.lo[1], b0.hi[1] };
        n[1] = F4{ b1.lo[0], b1.hi[0], b1.lo[1], b1.hi[1] };
        n[2] = F4{ b0.lo[2], b0.hi[2], b1.lo[2], b1.hi[2] };
        n[3] = F4{ bits_f((uint32_t)r0), bits_f((uint32_t)r1), 0.f, 0.f };
    };
    auto child_ref = [&](int32_t ch, Box &box) -> int32_t {
        const BuildNode &n = B.nodes[ch];
        box = padded(n.box);
        return n.leaf ? leaf_of(n) : dev_index[ch];
    };
    if (synth_root) {
        // The kernel always starts at an inner node.  A scene that is a single leaf gets a root whose
        // two children split that leaf (a 1-triangle scene lists the triangle twice: harmless, the
        // tie-break keeps the first); an empty scene gets one never-matching record (mask 0).
        if (ntris == 0) {
            out.tris.assign(3, F4{ 0, 0, 0, 0 });
            Box z; const float o[3] = { 0, 0, 0 }; z.grow(o);
            put(0, z, ~(int32_t)0, z, ~(int32_t)0);   // (record 0, one triangle)
        } else {
            const Box b = padded(B.nodes[root].box);
            const uint32_t n0 = (n_prims + 1) / 2, n1 = n_prims - n0;
            put(0, b, leaf_ref(0, n0), b, n1 ? leaf_ref(n0, n1) : leaf_ref(0, n0));
        }
    } else {
        for (size_t n = 0; n < B.nodes.size(); ++n) {
            if (dev_index[n] < 0) continue;
            Box b0, b1;
            const int32_t r0 = child_ref(B.nodes[n].child[0], b0), r1 = child_ref(B.nodes[n].child[1], b1);
            put((uint32_t)dev_index[n], b0, r0, b1, r1);
        }
    }

    out.boxes.clear();
    std::vector<int32_t> box_node_of_out;   // per entry of out.boxes: the build node that is its subtree's root
    for (uint32_t i = 0; i < n_prims && !synth_root; ++i) {
        const uint32_t b = B.prims[i].box;
        if (b == ~0u || box_root_node[b] < 0 || dev_index[(size_t)box_root_node[b]] < 0) continue;
        BoxLeaf bl;
        box_node_of_out.push_back(box_root_node[b]);
        bl.node = (uint32_t)dev_index[(size_t)box_root_node[b]];
        bl.first_rec = rec_first[i];
        for (int a = 0; a < 3; ++a) { bl.center[a] = (float)box_geom[b].c[a]; for (int k = 0; k < 3; ++k) bl.axis[k][a] = (float)box_geom[b].a[k][a]; }
        bl.mask = masks[B.prims[i].id];
        for (int f = 0; f < 6; ++f) {
            bl.face_rec[f] = 0; bl.face_bits[f] = 0;
            for (uint32_t j = 0; j < 6; ++j) {
                if (box_quads[b][j].q != box_geom[b].face_quad[f]) continue;
                bl.face_rec[f] = (uint8_t)(2u * j);
                // the quad's own parameters in the box's frame: X.e1 and Y.e2 run along the two axes other than the face's
                const uint32_t x = box_quads[b][j].x, y = box_quads[b][j].y;
                const int kx = f / 2, ia = kx == 0 ? 1 : 0, ja = kx == 2 ? 1 : 2;   // i < j, both != kx
                double l1[3] = { 0, 0, 0 }, l2[3] = { 0, 0, 0 };
                for (int ax = 0; ax < 3; ++ax)
                    for (int a = 0; a < 3; ++a) {
                        l1[ax] += box_geom[b].a[ax][a] * ((double)vtx(x, 1)[a] - (double)vtx(x, 0)[a]);
                        l2[ax] += box_geom[b].a[ax][a] * ((double)vtx(y, 2)[a] - (double)vtx(y, 0)[a]);
                    }
                const bool swap = std::fabs(l1[ja]) > std::fabs(l1[ia]);                // s runs along l_j
                const double ds = swap ? l1[ja] : l1[ia], dt = swap ? l2[ia] : l2[ja];
                bl.face_bits[f] = (uint8_t)((swap ? 1u : 0u) | (ds < 0 ? 2u : 0u) | (dt < 0 ? 4u : 0u));
            }
        }
        out.boxes.push_back(bl);
    }

    // ---- 4-wide collapse (used for scenes that stay in HBM): each BVH2 inner node pulls up grandchildren,
    //      largest surface area first, until it has four children.  Same leaves, same triangle records. ----
    constexpr int32_t kEmpty = (int32_t)0x80000000;
    struct Wide { int32_t child[4]; Box box[4]; };
    // Two flavours of the same collapse: the plain one (what the strict build and the tests see), and -- if the scene has boxes -- one in which the
    // root of a box's subtree is a LEAF whose code carries the index of its BoxLeaf (the shipped build's HBM traversal, trav_step_wide)
    std::vector<int32_t> box_of_node(B.nodes.size(), -1);
    for (size_t b = 0; b < out.boxes.size(); ++b) box_of_node[(size_t)box_node_of_out[b]] = (int32_t)b;
    out.n_boxes_real = (uint32_t)out.boxes.size();
    // ... and in that flavour a LONE QUAD is a box too, one of no thickness (half thickness 2^-30 of its longer edge: the two slab planes round to
    // the quad's own plane), so that a scene made of quads and cubes -- the reference's Scene API makes nothing else -- never runs the triangle
    // half of the unified step: frame rows 0, 1 = the quad's own (s, t) scaled to [-1, 1], row 2 = the unit normal over the half thickness; every
    // face names the quad's X record.  (BoxLeaf::node = ~0u: nothing the LDS layout could match.)
    if (!out.boxes.empty() && !synth_root) {
        for (size_t bn = 0; bn < B.nodes.size(); ++bn) {
            const BuildNode &n = B.nodes[bn];
            if (!n.leaf || n.count != 1u || n.boxprim != ~0u || dev_index.size() <= bn) continue;
            const Prim &pr = B.prims[n.first];
            if (pr.id2 == ~0u || pr.box != ~0u) continue;
            const uint32_t r0 = rec_first[n.first];
            const F4 *X = &out.tris[(size_t)r0 * 3], *Y = &out.tris[(size_t)(r0 + 1) * 3];
            const double p0[3] = { X[0].x, X[0].y, X[0].z }, e1[3] = { X[1].x, X[1].y, X[1].z }, e2[3] = { Y[2].x, Y[2].y, Y[2].z };
            auto cross = [](const double *u, const double *v, double *o) { o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0]; };
            auto dot = [](const double *u, const double *v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; };
            double nn[3], c1[3], c2[3];
            cross(e1, e2, nn);
            const double len = std::sqrt(dot(nn, nn)), size = std::sqrt(std::max(dot(e1, e1), dot(e2, e2)));
            if (!(len > 0.0) || !std::isfinite(len) || !(size > 0.0)) continue;
            for (int a = 0; a < 3; ++a) nn[a] /= len;
            cross(e2, nn, c1); cross(nn, e1, c2);
            const double s1 = dot(e1, c1), s2 = dot(e2, c2), eh = size * (1.0 / 1073741824.0);   // 2^-30: in fp32 the slab's two planes ARE the quad's plane (-lo_2 -+ 1 rounds to -lo_2), so t is the plane test's t
            if (s1 == 0.0 || s2 == 0.0) continue;
            BoxLeaf bl;
            bl.node = ~0u; bl.first_rec = r0; bl.mask = masks[pr.id];
            for (int a = 0; a < 3; ++a) {
                bl.center[a] = (float)(p0[a] + 0.5 * (e1[a] + e2[a]));
                bl.axis[0][a] = (float)(2.0 * c1[a] / s1); bl.axis[1][a] = (float)(2.0 * c2[a] / s2); bl.axis[2][a] = (float)(nn[a] / eh);
            }
            bool finite = true;
            for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(bl.axis[k][a]);
            if (!finite) continue;
            for (int f = 0; f < 6; ++f) { bl.face_rec[f] = 0; bl.face_bits[f] = 0; }
            box_of_node[bn] = (int32_t)out.boxes.size();
            out.boxes.push_back(bl);
        }
    }
    out.nodes4q_box.clear(); out.n_nodes4_box = 0; out.depth4_box = 0;
    const bool root_is_box = !synth_root && root >= 0 && box_of_node[(size_t)root] >= 0;
    for (int flavour = 0; flavour < ((out.boxes.empty() || synth_root || root_is_box) ? 1 : 2); ++flavour) {
    const bool box_flavour = flavour == 1;
    auto leaf_like = [&](int32_t bn) { return B.nodes[(size_t)bn].leaf || (box_flavour && box_of_node[(size_t)bn] >= 0); };
    std::vector<Wide> wide;
    uint32_t wdepth = 0;
    if (synth_root) {
        Wide w;
        for (int k = 0; k < 4; ++k) { w.child[k] = kEmpty; w.box[k] = Box(); }
        // reuse the two leaf references the BVH2 root got
        const F4 *n = &out.nodes[0];
        for (int k = 0; k < 2; ++k) {
            uint32_t bits; std::memcpy(&bits, k == 0 ? &n[3].x : &n[3].y, 4);
            w.child[k] = (int32_t)bits;
        }
        w.box[0].lo[0] = n[0].x; w.box[0].hi[0] = n[0].y; w.box[0].lo[1] = n[0].z; w.box[0].hi[1] = n[0].w; w.box[0].lo[2] = n[2].x; w.box[0].hi[2] = n[2].y;
        w.box[1].lo[0] = n[1].x; w.box[1].hi[0] = n[1].y; w.box[1].lo[1] = n[1].z; w.box[1].hi[1] = n[1].w; w.box[1].lo[2] = n[2].z; w.box[1].hi[2] = n[2].w;
        wide.push_back(w);
        wdepth = 1;
    } else {
        // iterative DFS; a work item = (build node, wide index it was assigned, depth)
        struct Item { int32_t bn; uint32_t wi; uint32_t d; };
        std::vector<Item> st;
        wide.emplace_back();
        st.push_back({ root, 0u, 1u });
        while (!st.empty()) {
            const Item it = st.back(); st.pop_back();
            wdepth = std::max(wdepth, it.d);
            int32_t ch[4]; int nch = 2;
            ch[0] = B.nodes[it.bn].child[0]; ch[1] = B.nodes[it.bn].child[1];
            while (nch < 4) {
                int best = -1; float best_area = -1.f;
                for (int k = 0; k < nch; ++k)
                    if (!leaf_like(ch[k])) { const float a = B.nodes[ch[k]].box.half_area(); if (a > best_area) { best_area = a; best = k; } }
                if (best < 0) break;
                const int32_t open = ch[best];
                ch[best] = B.nodes[open].child[0];
                ch[nch++] = B.nodes[open].child[1];
            }
            Wide w;
            for (int k = 0; k < 4; ++k) { w.child[k] = kEmpty; w.box[k] = Box(); }
            for (int k = 0; k < nch; ++k) {
                const BuildNode &c = B.nodes[ch[k]];
                w.box[k] = padded(c.box);
                if (box_flavour && box_of_node[(size_t)ch[k]] >= 0) {
                    // (the index of its BoxLeaf BEHIND the leaf records: the traversal addresses box b like record nrec + b, bvh_build.h)
                    w.child[k] = ~(int32_t)((((uint32_t)(out.tris.size() / 3) + (uint32_t)box_of_node[(size_t)ch[k]]) << 3) | kLeafBox);
                } else if (c.leaf) {
                    w.child[k] = leaf_of(c);
                } else {
                    w.child[k] = (int32_t)wide.size();
                    wide.emplace_back();
                    st.push_back({ ch[k], (uint32_t)w.child[k], it.d + 1 });
                }
            }
            wide[it.wi] = w;
        }
    }
    std::vector<F4> nodes4f((size_t)wide.size() * 8, F4{ 0, 0, 0, 0 });
    for (size_t i = 0; i < wide.size(); ++i) {
        const Wide &w = wide[i];
        F4 *n = &nodes4f[i * 8];
        float v[6][4];
        for (int k = 0; k < 4; ++k) {
            const bool e = w.child[k] == kEmpty;
            for (int a = 0; a < 3; ++a) { v[a * 2][k] = e ? 0.f : w.box[k].lo[a]; v[a * 2 + 1][k] = e ? 0.f : w.box[k].hi[a]; }
        }
        for (int r = 0; r < 6; ++r) n[r] = F4{ v[r][0], v[r][1], v[r][2], v[r][3] };
        n[6] = F4{ bits_f((uint32_t)w.child[0]), bits_f((uint32_t)w.child[1]), bits_f((uint32_t)w.child[2]), bits_f((uint32_t)w.child[3]) };
        n[7] = F4{ 0.f, 0.f, 0.f, 0.f };
    }

    std::vector<uint32_t> nodes4q((size_t)wide.size() * 16, 0u);
    for (size_t i = 0; i < wide.size(); ++i) quantize_node4(&nodes4f[i * 8].x, &nodes4q[i * 16]);
    if (!box_flavour) { out.n_nodes4 = (uint32_t)wide.size(); out.depth4 = wdepth; out.nodes4.swap(nodes4f); out.nodes4q.swap(nodes4q); }
    else { out.n_nodes4_box = (uint32_t)wide.size(); out.depth4_box = wdepth; out.nodes4q_box.swap(nodes4q); }
    }   // (flavours)

    // ---- 8-wide compressed collapse (q8node.h; TRG_WIDE8 builds) ----
    out.nodes8.clear(); out.rec8.clear(); out.rec8_flags.clear(); out.n_nodes8 = 0; out.depth8 = 0; out.wide8_ok = false;
    if (want_wide8) {
        struct Item { int32_t bn; uint32_t first, count; Box box; bool leaf; };   // a child: an inner build node, or a leaf over prims [first, first + count)
        auto item_of = [&](int32_t bn) { const BuildNode &n = B.nodes[bn]; return Item{ bn, n.first, n.count, n.box, n.leaf }; };
        struct QNode { std::vector<Item> ch; uint32_t depth; };
        std::vector<QNode> queue;   // breadth first: index = node index
        bool ok = true;
        {
            QNode r; r.depth = 1;
            if (synth_root) {
                if (ntris == 0) {
                    Box z; const float o[3] = { 0, 0, 0 }; z.grow(o);
                    r.ch.push_back(Item{ -1, 0u, 0u, z, true });     // (count 0: the never-matching record)
                } else {
                    const uint32_t n0 = (n_prims + 1) / 2, n1 = n_prims - n0;
                    r.ch.push_back(Item{ -1, 0u, n0, B.nodes[root].box, true });
                    if (n1) r.ch.push_back(Item{ -1, n0, n1, B.nodes[root].box, true });
                }
            } else {
                r.ch.push_back(item_of(B.nodes[root].child[0])); r.ch.push_back(item_of(B.nodes[root].child[1]));
            }
            queue.push_back(r);
        }
        for (size_t qi = 0; qi < queue.size() && ok; ++qi) {
            // open the inner child with the largest surface area until there are eight children (or nothing left to open)
            std::vector<Item> ch = queue[qi].ch;
            while (ch.size() < 8) {
                int best = -1; float best_area = -1.f;
                for (size_t k = 0; k < ch.size(); ++k)
                    if (!ch[k].leaf) { const float a = ch[k].box.half_area(); if (a > best_area) { best_area = a; best = (int)k; } }
                if (best < 0) break;
                const int32_t open = ch[best].bn;
                ch[best] = item_of(B.nodes[open].child[0]);
                ch.push_back(item_of(B.nodes[open].child[1]));
            }
            // slots: child c goes where dot(centre(c) - centre(node), (+-1, +-1, +-1)_slot) is largest, greedily over all (child, free slot) pairs
            Box nb;
            for (const Item &c : ch) nb.grow(c.box);
            float cn[3];
            for (int a = 0; a < 3; ++a) cn[a] = 0.5f * (nb.lo[a] + nb.hi[a]);
            int slot_of[8], child_in[8];
            for (int k = 0; k < 8; ++k) { slot_of[k] = -1; child_in[k] = -1; }
            for (size_t round = 0; round < ch.size(); ++round) {
                int bc = -1, bs = -1; float bv = -kInf;
                for (size_t c = 0; c < ch.size(); ++c) {
                    if (slot_of[c] >= 0) continue;
                    for (int sl = 0; sl < 8; ++sl) {
                        if (child_in[sl] >= 0) continue;
                        float v = 0.f;
                        for (int a = 0; a < 3; ++a) v += (0.5f * (ch[c].box.lo[a] + ch[c].box.hi[a]) - cn[a]) * (((sl >> a) & 1) ? 1.f : -1.f);
                        if (v > bv) { bv = v; bc = (int)c; bs = sl; }
                    }
                }
                slot_of[bc] = bs; child_in[bs] = bc;
            }
            // the node: inner children in slot order get consecutive indices, leaves in slot order consecutive record pairs
            uint32_t w[kQ8NodeDwords] = { 0 };
            uint32_t imask = 0, lmask = 0;
            const uint32_t child_base = (uint32_t)queue.size(), rec_base = (uint32_t)out.rec8.size();
            Box pb[8]; bool used[8];
            for (int sl = 0; sl < 8; ++sl) {
                used[sl] = child_in[sl] >= 0;
                if (!used[sl]) continue;
                const Item &c = ch[child_in[sl]];
                pb[sl] = padded(c.box);
                if (!c.leaf) {
                    imask |= 1u << sl;
                    QNode q; q.depth = queue[qi].depth + 1;
                    q.ch.push_back(item_of(B.nodes[c.bn].child[0])); q.ch.push_back(item_of(B.nodes[c.bn].child[1]));
                    queue.push_back(q);
                } else {
                    lmask |= 1u << sl;
                    const uint32_t r0 = c.count ? rec_first[c.first] : 0u, nrec = c.count ? rec_first[c.first + c.count] - r0 : 1u;
                    if (nrec > 2u) { ok = false; break; }            // (TRG_BVH_MAXLEAF > 2: this layout gives a leaf two record entries)
                    const bool q = c.count == 1u && B.prims[c.first].id2 != ~0u;
                    out.rec8.push_back(r0); out.rec8_flags.push_back((uint8_t)((nrec == 2u ? kRec8HasNext : 0u) | (q ? kRec8Quad : 0u)));
                    out.rec8.push_back(nrec == 2u ? r0 + 1u : ~0u); out.rec8_flags.push_back(0);
                }
            }
            if (!ok) break;
            out.depth8 = std::max(out.depth8, queue[qi].depth);
            // quantise against the union of the (padded) child boxes; per axis a power-of-two scale >= extent / 255
            uint32_t qlo[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } }, qhi[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
            float origin[3], scale[3];
            for (int a = 0; a < 3; ++a) {
                float mn = kInf, mx = -kInf;
                for (int sl = 0; sl < 8; ++sl) if (used[sl]) { mn = std::min(mn, pb[sl].lo[a]); mx = std::max(mx, pb[sl].hi[a]); }
                if (!(mn <= mx)) { mn = 0.f; mx = 0.f; }
                const double ext = (double)mx - (double)mn;
                int e = -100;
                if (ext > 0.0) { (void)std::frexp(ext / 255.0, &e); if (e < -100) e = -100; }
                for (;;) {
                    const double sc = std::ldexp(1.0, e);
                    bool fits = true;
                    uint32_t pl[2] = { 0, 0 }, ph[2] = { 0, 0 };
                    for (int sl = 0; sl < 8; ++sl) {
                        uint32_t l = 255u, h = 0u;   // an empty slot: the inverted box, never entered
                        if (used[sl]) {
                            const double dl = std::floor(((double)pb[sl].lo[a] - (double)mn) / sc), dh = std::ceil(((double)pb[sl].hi[a] - (double)mn) / sc);
                            if (dl < 0.0 || dh > 255.0 || !(dl <= 255.0) || !(dh >= 0.0)) { fits = false; break; }
                            l = (uint32_t)dl; h = (uint32_t)dh;
                        }
                        pl[sl >> 2] |= l << (8 * (sl & 3)); ph[sl >> 2] |= h << (8 * (sl & 3));
                    }
                    if (fits || e >= 127) {
                        if (!fits) for (int sl = 0; sl < 8; ++sl) { pl[sl >> 2] &= ~(255u << (8 * (sl & 3))); ph[sl >> 2] &= ~(255u << (8 * (sl & 3)));
                                                                     pl[sl >> 2] |= (used[sl] ? 0u : 255u) << (8 * (sl & 3)); ph[sl >> 2] |= (used[sl] ? 255u : 0u) << (8 * (sl & 3)); }
                        qlo[a][0] = pl[0]; qlo[a][1] = pl[1]; qhi[a][0] = ph[0]; qhi[a][1] = ph[1];
                        origin[a] = mn; scale[a] = (float)sc;
                        break;
                    }
                    ++e;
                }
            }
            auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
            w[0] = fbits(origin[0]); w[1] = fbits(origin[1]); w[2] = fbits(origin[2]); w[3] = fbits(scale[0]);
            w[4] = child_base; w[5] = rec_base; w[6] = imask | (lmask << 8);
            w[7] = (fbits(scale[1]) >> 16) | (fbits(scale[2]) & 0xFFFF0000u);
            w[8] = qlo[0][0]; w[9] = qlo[0][1]; w[10] = qlo[1][0]; w[11] = qlo[1][1];
            w[12] = qlo[2][0]; w[13] = qlo[2][1]; w[14] = qhi[0][0]; w[15] = qhi[0][1];
            w[16] = qhi[1][0]; w[17] = qhi[1][1]; w[18] = qhi[2][0]; w[19] = qhi[2][1];
            out.nodes8.insert(out.nodes8.end(), w, w + kQ8NodeDwords);
        }
        if (ok) { out.n_nodes8 = (uint32_t)queue.size(); out.wide8_ok = true; }
        else { out.nodes8.clear(); out.rec8.clear(); out.rec8_flags.clear(); out.depth8 = 0; }
    }

    // SAH cost (reporting only)
    double cost = 0.0;
    if (!synth_root) {
        const double ra = B.nodes[root].box.half_area();
        if (ra > 0)
            for (const BuildNode &n : B.nodes) cost += (n.leaf ? (double)n.count : 1.2) * n.box.half_area() / ra;
    }
    out.sah_cost = cost;
}

}  // namespace trg
