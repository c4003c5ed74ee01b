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
