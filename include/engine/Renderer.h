// engine/Renderer.h -- toyraygun::Renderer, the plugin surface a backend implements
// (reference src/engine/Renderer.h:21-57, Renderer.cpp:7-89).  Call order the app relies on
// (main.cpp:21-95): Engine::init -> createRenderer()->init() -> addShader x3 -> setCameraPosition /
// setCameraLookAt -> loadScene(scene) -> loop renderFrame().
#pragma once
#include <bx/math.h>

#include <string>
#include <vector>

#include "engine/Engine.h"
#include "engine/Scene.h"
#include "engine/Shader.h"
#include "engine/Uniforms.h"

#define MATERIAL_DEFAULT 1
#define MATERIAL_EMISSIVE 2

namespace toyraygun {

class Renderer {
public:
    Renderer();
    virtual ~Renderer() {}

    virtual bool init();                  // base: reads Engine width/height, returns false (Renderer.cpp:18-27)
    virtual void destroy();
    virtual void loadScene(Scene *scene);  // borrows `scene` for the call only
    virtual void renderFrame();           // base: ++m_frameIndex

    void getViewProjMtx(float *mtxOut);
    bx::Vec3 getCameraPosition();
    void setCameraPosition(bx::Vec3 position);
    void setCameraLookAt(bx::Vec3 position);
    void updateCamera();

    virtual void addShader(Shader *shader);
    virtual Shader *getShader(std::string path);

protected:
    int m_frameIndex;
    std::vector<Shader *> m_shaders;
    int m_width, m_height;
    float m_aspectRatio;
    bx::Vec3 m_eye, m_up, m_at;
    float m_viewMtx[16], m_projMtx[16], m_viewProjMtx[16];
};

}  // namespace toyraygun
