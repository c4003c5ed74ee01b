#include "bvh_build.h"
#include <cstdio>
#include <random>
#include <vector>
int main() {
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    for (uint32_t n : { 0u, 1u, 2u, 3u, 36u, 257u, 5000u, 200000u }) {
        std::vector<float> pos((size_t)n * 9);
        std::vector<uint32_t> idx((size_t)n * 3), mat(n, 1u);
        for (uint32_t t = 0; t < n; ++t) {
            float c[3] = { U(rng), U(rng), U(rng) };
            for (int v = 0; v < 3; ++v) for (int a = 0; a < 3; ++a) pos[(size_t)t * 9 + v * 3 + a] = c[a] + 0.05f * U(rng) * (t % 7 == 0 ? 0.f : 1.f);
            for (int v = 0; v < 3; ++v) idx[(size_t)t * 3 + v] = t * 3 + v;
        }
        trg::Bvh b;
        trg::build_bvh(pos.data(), idx.data(), mat.data(), n, b);
        std::printf("n=%u nodes=%u nodes4=%u depth=%u depth4=%u tris=%zu q=%zu\n", n, b.n_nodes, b.n_nodes4, b.depth, b.depth4, b.tris.size() / 3, b.nodes4q.size() / 16);
    }
    return 0;
}
