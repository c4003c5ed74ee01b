// main_cornell.cpp -- headless demo following the reference app's call order (reference src/main.cpp:16-98):
// engine init -> three shaders -> renderer init -> camera -> Cornell box -> frame loop; then writes a PNG.
//   toyraygun_cornell [width height frames bounces out.png]
#include <stdio.h>
#include <stdlib.h>

#include <iostream>

#include "cornellBox.h"
#include "engine/Engine.h"
#include "engine/HipRenderer.h"
#include "engine/Renderer.h"
#include "engine/Shader.h"

using namespace toyraygun;

static Shader *loadShader(const char *name, ShaderType type, const char *const *fns, const ShaderFunctionType *types, int n) {
    Shader *s = Engine::createShader();
    if (!s->load(name)) { std::cout << "Failed to load " << name << " shader." << std::endl; return nullptr; }
    for (int i = 0; i < n; ++i) s->addFunction(fns[i], types[i]);
    if (!s->compile(type)) { std::cout << "Failed to compile " << name << " shader." << std::endl; return nullptr; }
    return s;
}

int main(int argc, char **argv) {
    const int width = argc > 1 ? atoi(argv[1]) : 1024, height = argc > 2 ? atoi(argv[2]) : 768;
    const int frames = argc > 3 ? atoi(argv[3]) : 64, bounces = argc > 4 ? atoi(argv[4]) : 3;
    const char *out = argc > 5 ? argv[5] : "cornell.png";

    Engine *engine = Engine::instance();
    engine->init(width, height);
    engine->setFrameBudget(frames);

    const char *rtFns[] = { "raygen", "primaryHit", "primaryMiss", "shadowHit", "shadowMiss" };
    const ShaderFunctionType rtTypes[] = { ShaderFunctionType::RayGen, ShaderFunctionType::ClosestHit, ShaderFunctionType::Miss,
                                           ShaderFunctionType::ShadowHit, ShaderFunctionType::ShadowMiss };
    const char *accFns[] = { "accumulate" };
    const ShaderFunctionType accTypes[] = { ShaderFunctionType::Compute };
    const char *ppFns[] = { "vert", "frag" };
    const ShaderFunctionType ppTypes[] = { ShaderFunctionType::Vertex, ShaderFunctionType::Fragment };
    Shader *rt = loadShader("Raytracing", ShaderType::Raytrace, rtFns, rtTypes, 5);
    Shader *acc = loadShader("Accumulate", ShaderType::Compute, accFns, accTypes, 1);
    Shader *pp = loadShader("PostProcessing", ShaderType::Graphics, ppFns, ppTypes, 2);
    if (!rt || !acc || !pp) return -1;

    Renderer *renderer = Engine::createRenderer();
    if (!renderer->init()) { std::cout << "Renderer failed to initialize." << std::endl; return -1; }
    renderer->addShader(rt);
    renderer->addShader(acc);
    renderer->addShader(pp);
    renderer->setCameraPosition(bx::Vec3(0.0f, 1.0f, 3.38f));
    renderer->setCameraLookAt(bx::Vec3(0.0f, 1.0f, -1.0f));

    Scene *scene = createCornellBoxScene();
    renderer->loadScene(scene);

    HipRenderer *hip = static_cast<HipRenderer *>(renderer);
    hip->setBounces((unsigned int)bounces);
    while (!engine->hasQuit()) {
        engine->pollEvents();
        if (engine->hasQuit()) break;
        renderer->renderFrame();
    }
    printf("%d frames, %llu rays, last frame %.3f ms\n", hip->getFrameIndex(), (unsigned long long)hip->getRayCount(), hip->getLastRenderMs());
    if (!hip->savePNG(out)) { std::cout << "Failed to write " << out << std::endl; return -1; }
    printf("wrote %s\n", out);
    return 0;
}
