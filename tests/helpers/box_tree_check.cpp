#include "bvh_build.h"
#include "q4node.h"
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
