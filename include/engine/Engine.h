// engine/Engine.h -- headless counterpart of toyraygun::Engine (reference src/engine/Engine.h:39-56,
// Engine.cpp:22-121).  Same singleton + factory surface; no SDL window: init() only records the
// framebuffer size, and the frame loop `while (!engine->hasQuit())` ends after a frame budget
// (setFrameBudget / env TOYRAYGUN_FRAMES) because there is no window to close.
#pragma once
#include <string>

namespace toyraygun {

class Shader;
class Renderer;

class Engine {
public:
    static Engine *instance();
    static void initPIXDebugger();  // reference Engine.cpp:123-180; nothing to load here

    static Shader *createShader();       // -> HipShader (Engine.cpp:32-41)
    static Renderer *createRenderer();   // -> HipRenderer (Engine.cpp:43-52)
    static std::string getRuntimeShaderPath();
    static std::string getRuntimeShaderExt();

    virtual void init(int width, int height);
    virtual void destroy();
    virtual int getWidth();
    virtual int getHeight();
    virtual bool hasQuit();
    virtual void pollEvents();

    // headless additions (no reference equivalent)
    void setFrameBudget(int frames);  // hasQuit() turns true after this many pollEvents(); <0: never
    void setDevice(int device);       // HIP device the next createRenderer()'s init() will use
    int getDevice() const { return m_device; }
    virtual ~Engine() {}

protected:
    Engine();
    static Engine *m_instance;
    int m_width, m_height;
    bool m_quit;
    int m_frameBudget, m_framesPolled, m_device;
};

}  // namespace toyraygun
