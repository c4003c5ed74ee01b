// cornellBox.h -- the demo scene of the reference app (reference src/cornellBox.h:11-52): two boxes,
// five walls and a ceiling light, 36 triangles, built through the public Scene API.  Object order,
// colours and bx::mtxSRT arguments are the reference's, so the five Scene vectors come out the same.
#pragma once
#include "engine/Scene.h"

namespace toyraygun_scenes {

struct Placement {
    char kind;  // 'c' cube, 'p' plane, 'l' area light
    float color[3];
    float scale[3], rot[3], pos[3];
};

inline const Placement *cornellPlacements(int *count) {
    static const float kWhite[3] = { 0.725f, 0.71f, 0.68f };
    static const Placement table[] = {
        { 'c', { kWhite[0], kWhite[1], kWhite[2] }, { 0.6f, 0.6f, 0.6f }, { 0.0f, 0.3f, 0.0f }, { 0.3275f, 0.3f, 0.3725f } },   // short box
        { 'c', { kWhite[0], kWhite[1], kWhite[2] }, { 0.6f, 1.2f, 0.6f }, { 0.0f, -0.3f, 0.0f }, { -0.335f, 0.6f, -0.29f } },  // tall box
        { 'p', { kWhite[0], kWhite[1], kWhite[2] }, { 2.0f, 2.0f, 2.0f }, { 0.0f, 0.0f, bx::kPi }, { 0.0f, 1.0f, 0.0f } },        // ceiling
        { 'p', { kWhite[0], kWhite[1], kWhite[2] }, { 2.0f, 2.0f, 2.0f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 1.0f, 0.0f } },           // floor
        { 'p', { 0.63f, 0.065f, 0.05f }, { 2.0f, 2.0f, 2.0f }, { 0.0f, 0.0f, bx::kPi / 2.0f }, { 0.0f, 1.0f, 0.0f } },            // left wall (red)
        { 'p', { 0.14f, 0.491f, 0.05f }, { 2.0f, 2.0f, 2.0f }, { 0.0f, 0.0f, -bx::kPi / 2.0f }, { 0.0f, 1.0f, 0.0f } },           // right wall (green)
        { 'p', { kWhite[0], kWhite[1], kWhite[2] }, { 2.0f, 2.0f, 2.0f }, { -bx::kPi / 2.0f, 0.0f, 0.0f }, { 0.0f, 1.0f, 0.0f } }, // back wall
        { 'l', { 1.0f, 1.0f, 1.0f }, { 0.5f, 1.98f, 0.5f }, { 0.0f, 0.0f, bx::kPi }, { 0.0f, 1.0f, 0.0f } },                     // light
    };
    *count = (int)(sizeof(table) / sizeof(table[0]));
    return table;
}

}  // namespace toyraygun_scenes

inline toyraygun::Scene *createCornellBoxScene() {
    toyraygun::Scene *scene = new toyraygun::Scene();
    int n = 0;
    const toyraygun_scenes::Placement *p = toyraygun_scenes::cornellPlacements(&n);
    float m[16];
    for (int i = 0; i < n; ++i) {
        bx::mtxSRT(m, p[i].scale[0], p[i].scale[1], p[i].scale[2], p[i].rot[0], p[i].rot[1], p[i].rot[2], p[i].pos[0],
                   p[i].pos[1], p[i].pos[2]);
        const bx::Vec3 color(p[i].color[0], p[i].color[1], p[i].color[2]);
        if (p[i].kind == 'c') scene->addCube(color, m);
        else if (p[i].kind == 'p') scene->addPlane(color, m);
        else scene->addAreaLight(color, m);
    }
    return scene;
}

// BASELINE config C4 (a synthetic this project defines): the Cornell box plus the reference's cube
// mesh (Scene::addCube) replicated on an n x n x n lattice inside the room.  Lattice region x,z in
// [-0.9, 0.9], y in [0.05, 1.85]; cube side 0.25 x cell; per-cube jitter, y-rotation and colour from
// pcg_hash32(seed ^ ...).  n = 44 -> 36 + 12 * 44^3 = 1,022,244 triangles.
namespace toyraygun_scenes {
inline uint32_t pcgHash32(uint32_t v) {
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
}  // namespace toyraygun_scenes

inline toyraygun::Scene *createCornellLatticeScene(int n, uint32_t seed = 0x5EED0002u) {
    using toyraygun_scenes::pcgHash32;
    toyraygun::Scene *scene = createCornellBoxScene();
    const float lo[3] = { -0.9f, 0.05f, -0.9f }, hi[3] = { 0.9f, 1.85f, 0.9f };
    float cell[3];
    for (int a = 0; a < 3; ++a) cell[a] = (hi[a] - lo[a]) / (float)n;
    float minCell = cell[0];
    if (cell[1] < minCell) minCell = cell[1];
    if (cell[2] < minCell) minCell = cell[2];
    const float side = 0.25f * minCell;
    const float unit = 1.0f / 16777216.0f;
    scene->m_vertexBuffer.reserve(scene->m_vertexBuffer.size() + (size_t)36 * n * n * n);
    scene->m_normalBuffer.reserve(scene->m_normalBuffer.size() + (size_t)36 * n * n * n);
    scene->m_colorBuffer.reserve(scene->m_colorBuffer.size() + (size_t)36 * n * n * n);
    scene->m_indexBuffer.reserve(scene->m_indexBuffer.size() + (size_t)36 * n * n * n);
    uint32_t id = 0;
    float m[16];
    for (int iz = 0; iz < n; ++iz)
        for (int iy = 0; iy < n; ++iy)
            for (int ix = 0; ix < n; ++ix, ++id) {
                const uint32_t h0 = pcgHash32(seed ^ (id * 4u + 0u)), h1 = pcgHash32(seed ^ (id * 4u + 1u));
                const uint32_t h2 = pcgHash32(seed ^ (id * 4u + 2u)), h3 = pcgHash32(seed ^ (id * 4u + 3u));
                const float j0 = (float)(h0 >> 8) * unit - 0.5f, j1 = (float)(h1 >> 8) * unit - 0.5f;
                const float j2 = (float)(h2 >> 8) * unit - 0.5f;
                const float rot = (float)(h3 >> 8) * unit * 1.5707964f;
                const float cx = lo[0] + ((float)ix + 0.5f + 0.3f * j0) * cell[0];
                const float cy = lo[1] + ((float)iy + 0.5f + 0.3f * j1) * cell[1];
                const float cz = lo[2] + ((float)iz + 0.5f + 0.3f * j2) * cell[2];
                const bx::Vec3 color(0.3f + 0.6f * (float)((h0 >> 3) & 31) / 31.0f, 0.3f + 0.6f * (float)((h1 >> 3) & 31) / 31.0f,
                                     0.3f + 0.6f * (float)((h2 >> 3) & 31) / 31.0f);
                bx::mtxSRT(m, side, side, side, 0.0f, rot, 0.0f, cx, cy, cz);
                scene->addCube(color, m);
            }
    return scene;
}
