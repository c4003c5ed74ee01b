#include "bvh_build.h"
#include "q4node.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
using namespace trg;
// a cube's 36 vertices (12 triangles, 6 quads, the (a,b,c)(a,c,d) pattern)
static void add_cube(std::vector<float> &pos, std::vector<uint32_t> &idx, std::vector<uint32_t> &mat, float cx, float cy, float cz, float h) {
    const float c[8][3] = { {-1,-1,-1},{1,-1,-1},{1,1,-1},{-1,1,-1},{-1,-1,1},{1,-1,1},{1,1,1},{-1,1,1} };
    const int f[6][4] = { {0,1,2,3},{5,4,7,6},{4,0,3,7},{1,5,6,2},{3,2,6,7},{4,5,1,0} };
    for (int q = 0; q < 6; ++q) {
        const int t[6] = { f[q][0], f[q][1], f[q][2], f[q][0], f[q][2], f[q][3] };
        for (int k = 0; k < 6; ++k) { idx.push_back((uint32_t)(pos.size() / 3)); pos.push_back(cx + h * c[t[k]][0]); pos.push_back(cy + h * c[t[k]][1]); pos.push_back(cz + h * c[t[k]][2]); }
        mat.push_back(1); mat.push_back(1);
    }
}
int main() {
    std::mt19937 rng(5);
    for (int n : { 1, 2, 7, 300, 9000 }) {
        std::vector<float> pos; std::vector<uint32_t> idx, mat;
        std::uniform_real_distribution<float> U(-5.f, 5.f);
        for (int i = 0; i < n; ++i) add_cube(pos, idx, mat, U(rng), U(rng), U(rng), 0.1f);
        // a lone triangle and two lone quads too (the quads become boxes of no thickness in the box flavour)
        for (int k = 0; k < 3; ++k) { idx.push_back((uint32_t)(pos.size() / 3)); pos.push_back(U(rng)); pos.push_back(U(rng)); pos.push_back(U(rng)); }
        mat.push_back(1);
        for (int q = 0; q < 2; ++q) {
            const float a[3] = { U(rng), U(rng), U(rng) }, e1[3] = { 1.f, 0.25f, 0.f }, e2[3] = { 0.f, 0.5f, 2.f };
            const float v[4][3] = { { a[0], a[1], a[2] }, { a[0] + e1[0], a[1] + e1[1], a[2] + e1[2] }, { a[0] + e1[0] + e2[0], a[1] + e1[1] + e2[1], a[2] + e1[2] + e2[2] }, { a[0] + e2[0], a[1] + e2[1], a[2] + e2[2] } };
            const int t[6] = { 0, 1, 2, 0, 2, 3 };
            for (int k = 0; k < 6; ++k) { idx.push_back((uint32_t)(pos.size() / 3)); for (int c = 0; c < 3; ++c) pos.push_back(v[t[k]][c]); }
            mat.push_back(1); mat.push_back(1);
        }
        Bvh b;
        build_bvh(pos.data(), idx.data(), mat.data(), (uint32_t)mat.size(), b, false, true);
        if (n == 7) {
            // hostile cubes: the same scene again with cubes of no size, of no thickness, at 1e30, with a NaN / an infinite corner, with two materials,
            // and one whose corner is off by 1e-3 -- none of them may be taken for a box (or crash the builder); the seven good ones still are
            std::vector<float> p2 = pos; std::vector<uint32_t> i2 = idx, m2 = mat;
            const size_t v0 = p2.size() / 3;
            auto cube_at = [&](float h, float cx) { add_cube(p2, i2, m2, cx, 8.f, 8.f, h); return p2.size() / 3 - 36; };
            cube_at(0.f, 1.f);                                                         // a point
            { const size_t c = cube_at(0.1f, 2.f); for (size_t v = c; v < c + 36; ++v) p2[v * 3 + 2] = 8.f; }          // flat
            { const size_t c = cube_at(0.1f, 3.f); for (size_t v = c; v < c + 36; ++v) for (int a = 0; a < 3; ++a) p2[v * 3 + a] *= 1e30f; }
            { const size_t c = cube_at(0.1f, 4.f); p2[c * 3] = std::nanf(""); }
            { const size_t c = cube_at(0.1f, 5.f); p2[(c + 7) * 3 + 1] = INFINITY; }
            { cube_at(0.1f, 6.f); m2[m2.size() - 1] = 2; }
            { const size_t c = cube_at(0.1f, 7.f); p2[(c + 2) * 3] += 1e-3f; }
            (void)v0;
            Bvh h;
            build_bvh(p2.data(), i2.data(), m2.data(), (uint32_t)m2.size(), h, false, true);
            if (h.n_boxes_real != 7u) { printf("hostile cubes: %u boxes\n", h.n_boxes_real); return 1; }
            printf("hostile cubes: none taken for a box\n");
        }
        const uint32_t nrec = (uint32_t)(b.tris.size() / 3);
        std::vector<int> seen(nrec, 0);
        size_t box_refs = 0;
        for (uint32_t i = 0; i < b.n_nodes4_box; ++i) {
            const uint32_t *w = &b.nodes4q_box[(size_t)i * 16];
            for (int k = 0; k < 4; ++k) {
                const int32_t c = (int32_t)w[12 + k];
                if (c == (int32_t)0x80000000) continue;
                if (c >= 0) { if ((uint32_t)c >= b.n_nodes4_box) { printf("bad inner %d\n", c); return 1; } continue; }
                const uint32_t code = (uint32_t)~c, first = code >> 3, cnt = code & 7u;
                if (cnt == kLeafBox) {
                    if (first < nrec || first - nrec >= b.boxes.size()) { printf("bad box %u\n", first); return 1; }
                    const uint32_t first = (code >> 3) - nrec;      // (a box is addressed like a record behind the records)
                    ++box_refs;
                    for (uint32_t r = 0; r < (b.boxes[first].node == ~0u ? 2u : 12u); ++r) seen[b.boxes[first].first_rec + r]++;
                } else {
                    const uint32_t m = leaf_count(code);
                    if (first + m > nrec) { printf("bad leaf\n"); return 1; }
                    for (uint32_t r = 0; r < m; ++r) seen[first + r]++;
                }
            }
        }
        bool ok = true;
        if (b.n_nodes4_box) for (uint32_t r = 0; r < nrec; ++r) ok = ok && seen[r] == 1;
        if (b.n_boxes_real != (uint32_t)n || (b.n_nodes4_box && b.boxes.size() != (size_t)n + 2)) { printf("box counts: %u real, %zu all\n", b.n_boxes_real, b.boxes.size()); return 1; }
        printf("n=%d tris=%zu boxes=%zu nodes4=%u nodes4_box=%u box_refs=%zu depth4 %u/%u every record once: %s\n", n, mat.size(), b.boxes.size(), b.n_nodes4, b.n_nodes4_box, box_refs, b.depth4, b.depth4_box, ok ? "yes" : "NO");
        if (!ok || box_refs != (b.n_nodes4_box ? b.boxes.size() : 0)) return 1;
    }
    return 0;
}
