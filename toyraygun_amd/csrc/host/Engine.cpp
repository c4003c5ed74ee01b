// Engine.cpp -- headless toyraygun::Engine (see include/engine/Engine.h).
#include "engine/Engine.h"

#include <stdlib.h>

#include "engine/HipRenderer.h"
#include "engine/Shader.h"

namespace toyraygun {

Engine *Engine::m_instance = nullptr;

Engine::Engine() : m_width(0), m_height(0), m_quit(false), m_frameBudget(64), m_framesPolled(0), m_device(0) {
    if (const char *env = getenv("TOYRAYGUN_FRAMES")) m_frameBudget = atoi(env);
    if (const char *env = getenv("TOYRAYGUN_DEVICE")) m_device = atoi(env);
}

Engine *Engine::instance() {
    if (!m_instance) m_instance = new Engine();
    return m_instance;
}

void Engine::initPIXDebugger() {}

Shader *Engine::createShader() { return new Shader(); }
Renderer *Engine::createRenderer() { return new HipRenderer(); }
std::string Engine::getRuntimeShaderPath() { return "shaders/hip/"; }
std::string Engine::getRuntimeShaderExt() { return "hip"; }

void Engine::init(int width, int height) {
    m_width = width;
    m_height = height;
    m_quit = false;
    m_framesPolled = 0;
}
void Engine::destroy() { m_quit = true; }
int Engine::getWidth() { return m_width; }
int Engine::getHeight() { return m_height; }
bool Engine::hasQuit() { return m_quit; }
void Engine::pollEvents() {
    // the reference drains SDL events here (Engine.cpp:109-121); headless: count frames
    if (m_frameBudget >= 0 && m_framesPolled >= m_frameBudget) m_quit = true;
    ++m_framesPolled;
}
void Engine::setFrameBudget(int frames) { m_frameBudget = frames; m_framesPolled = 0; m_quit = false; }
void Engine::setDevice(int device) { m_device = device; }

}  // namespace toyraygun
