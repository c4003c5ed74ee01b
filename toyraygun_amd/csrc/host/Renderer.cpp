// Renderer.cpp -- toyraygun::Renderer base class: camera + shader registry (a2 of SURVEY 8a;
// reference src/engine/Renderer.cpp:7-89).
#include "engine/Renderer.h"

#include <string.h>

namespace toyraygun {

Renderer::Renderer()
    : m_frameIndex(0), m_width(0), m_height(0), m_aspectRatio(1.0f), m_eye(0.0f, 0.0f, 0.0f), m_up(0.0f, 1.0f, 0.0f),
      m_at(0.0f, 0.0f, 0.0f) {
    bx::mtxIdentity(m_viewMtx);
    bx::mtxIdentity(m_projMtx);
    bx::mtxIdentity(m_viewProjMtx);
}

bool Renderer::init() {
    Engine *engine = Engine::instance();
    m_width = engine->getWidth();
    m_height = engine->getHeight();
    m_aspectRatio = float(m_width) / float(m_height);
    return false;  // the base class is not a usable backend (Renderer.cpp:26)
}
void Renderer::destroy() {}
void Renderer::loadScene(Scene *) {}
void Renderer::renderFrame() { m_frameIndex++; }

void Renderer::addShader(Shader *shader) { m_shaders.push_back(shader); }
Shader *Renderer::getShader(std::string path) {
    for (size_t i = 0; i < m_shaders.size(); ++i)
        if (m_shaders[i]->m_path == path) return m_shaders[i];
    return nullptr;
}

void Renderer::getViewProjMtx(float *mtxOut) { memcpy(mtxOut, m_viewProjMtx, sizeof(m_viewProjMtx)); }
bx::Vec3 Renderer::getCameraPosition() { return m_eye; }
void Renderer::setCameraPosition(bx::Vec3 position) { m_eye = position; updateCamera(); }
void Renderer::setCameraLookAt(bx::Vec3 position) { m_at = position; updateCamera(); }

// view = lookAt(eye, at, up, RH); proj = 45 deg, near 1, far 125, homogeneous NDC, RH (Renderer.cpp:84-89)
void Renderer::updateCamera() {
    bx::mtxLookAt(m_viewMtx, m_eye, m_at, m_up, bx::Handness::Right);
    bx::mtxProj(m_projMtx, 45.0f, m_aspectRatio, 1.0f, 125.0f, true, bx::Handness::Right);
    bx::mtxMul(m_viewProjMtx, m_viewMtx, m_projMtx);
}

}  // namespace toyraygun
