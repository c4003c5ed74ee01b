// capi_host.cpp -- flat C entry points over the C++ plugin surface, so the Python tests can drive
// Engine / Renderer / Scene / HipRenderer exactly the way the reference app does (main.cpp:16-98).
// Test plumbing for libtoyraygun.so; the product boundary is include/trg.h.
#include <string.h>

#include <chrono>

#include "cornellBox.h"
#include "engine/Engine.h"
#include "engine/HipRenderer.h"
#include "engine/Renderer.h"
#include "engine/Shader.h"
#include "engine/Texture.h"

using namespace toyraygun;

extern "C" {

// ---- scene (a1) ----
void *trh_scene_new() { return new Scene(); }
void *trh_scene_cornell() { return createCornellBoxScene(); }
void *trh_scene_lattice(int n, uint32_t seed) { return createCornellLatticeScene(n, seed); }
void trh_scene_free(void *s) { delete static_cast<Scene *>(s); }
void trh_scene_add(void *s, int kind, const float *color3, const float *mtx16) {
    Scene *sc = static_cast<Scene *>(s);
    float m[16];
    memcpy(m, mtx16, sizeof(m));
    const bx::Vec3 c(color3[0], color3[1], color3[2]);
    if (kind == 0) sc->addCube(c, m);
    else if (kind == 1) sc->addPlane(c, m);
    else sc->addAreaLight(c, m);
}
void trh_scene_add_mesh(void *s, const float *verts3, const float *normals3, const uint32_t *indices, int triCount,
                        const float *mtx16, const float *color3, unsigned int materialID) {
    float m[16];
    memcpy(m, mtx16, sizeof(m));
    static_cast<Scene *>(s)->addMesh(reinterpret_cast<const bx::Vec3 *>(verts3), reinterpret_cast<const bx::Vec3 *>(normals3),
                                     indices, triCount, m, bx::Vec3(color3[0], color3[1], color3[2]), materialID);
}
void trh_scene_add_mesh_colors(void *s, const float *verts3, const float *normals3, const float *colors3, const uint32_t *indices,
                               int triCount, const float *mtx16, unsigned int materialID) {
    float m[16];
    memcpy(m, mtx16, sizeof(m));
    static_cast<Scene *>(s)->addMesh(reinterpret_cast<const bx::Vec3 *>(verts3), reinterpret_cast<const bx::Vec3 *>(normals3),
                                     reinterpret_cast<const bx::Vec3 *>(colors3), indices, triCount, m, materialID);
}
// ---- textures (N4) ----
void *trh_texture_load(const char *path) {
    Texture *t = new Texture;
    if (!t->loadFile(path)) { delete t; return nullptr; }
    return t;
}
void *trh_texture_from_rgba(const uint8_t *rgba, int w, int h) {
    Texture *t = new Texture;
    t->init(w, h, 4);
    memcpy(t->getBufferPointer(), rgba, (size_t)w * h * 4);
    return t;
}
void trh_texture_info(void *t, int *w, int *h, int *channels) {
    Texture *x = static_cast<Texture *>(t);
    *w = x->getWidth(); *h = x->getHeight(); *channels = x->getChannels();
}
void trh_texture_copy(void *t, uint8_t *out) {
    Texture *x = static_cast<Texture *>(t);
    memcpy(out, x->getBufferPointer(), x->getBufferSize());
}
void trh_texture_free(void *t) {
    Texture *x = static_cast<Texture *>(t);
    x->destroy();
    delete x;
}
void trh_scene_add_textured_mesh(void *s, const float *verts3, const float *normals3, const float *uvs2, const uint32_t *indices, int triCount,
                                 const float *mtx16, const float *color3, unsigned int materialID, void *texture) {
    float m[16];
    memcpy(m, mtx16, sizeof(m));
    static_cast<Scene *>(s)->addMesh(reinterpret_cast<const bx::Vec3 *>(verts3), reinterpret_cast<const bx::Vec3 *>(normals3), uvs2, indices,
                                     triCount, m, bx::Vec3(color3[0], color3[1], color3[2]), materialID, static_cast<Texture *>(texture));
}
int trh_scene_add_obj_textured(void *s, const char *path, const float *mtx16, const float *color3, unsigned int materialID, void *texture) {
    float m[16];
    memcpy(m, mtx16, sizeof(m));
    return static_cast<Scene *>(s)->addObj(path, m, bx::Vec3(color3[0], color3[1], color3[2]), materialID, static_cast<Texture *>(texture));
}
// uv [nVerts*2] and texture ids [nTris], padded with "no texture" for triangles added after the last textured mesh
void trh_scene_copy_textures(void *s, float *uv, uint32_t *ids) {
    Scene *sc = static_cast<Scene *>(s);
    const size_t nv = sc->m_vertexBuffer.size(), nt = sc->m_materialIDBuffer.size();
    for (size_t i = 0; i < nv * 2; ++i) uv[i] = i < sc->m_uvBuffer.size() ? sc->m_uvBuffer[i] : 0.0f;
    for (size_t i = 0; i < nt; ++i) ids[i] = i < sc->m_textureIDBuffer.size() ? sc->m_textureIDBuffer[i] : 0u;
}
// the textures the scene actually registered (Scene::textureID: on first use by a face with texture coordinates), in id order:
// out[k] = the Texture of id k + 1.  Returns the count; out may be NULL / cap 0 to query it.
unsigned int trh_scene_texture_handles(void *s, void **out, unsigned int cap) {
    Scene *sc = static_cast<Scene *>(s);
    for (size_t k = 0; k < sc->m_textures.size() && k < cap && out; ++k) out[k] = sc->m_textures[k];
    return (unsigned int)sc->m_textures.size();
}
int trh_scene_add_obj(void *s, const char *path, const float *mtx16, const float *color3, unsigned int materialID) {
    float m[16];
    memcpy(m, mtx16, sizeof(m));
    return static_cast<Scene *>(s)->addObj(path, m, bx::Vec3(color3[0], color3[1], color3[2]), materialID);
}
unsigned int trh_scene_counts(void *s, unsigned int *nTris) {
    Scene *sc = static_cast<Scene *>(s);
    if (nTris) *nTris = (unsigned int)sc->m_materialIDBuffer.size();
    return (unsigned int)sc->m_vertexBuffer.size();
}
void trh_scene_copy(void *s, float *pos, float *nrm, float *col, uint32_t *idx, uint32_t *mat) {
    Scene *sc = static_cast<Scene *>(s);
    const size_t nv = sc->m_vertexBuffer.size(), nt = sc->m_materialIDBuffer.size();
    if (nv) {
        memcpy(pos, &sc->m_vertexBuffer[0], nv * 12);
        memcpy(nrm, &sc->m_normalBuffer[0], nv * 12);
        memcpy(col, &sc->m_colorBuffer[0], nv * 12);
        memcpy(idx, &sc->m_indexBuffer[0], nv * 4);
    }
    if (nt) memcpy(mat, &sc->m_materialIDBuffer[0], nt * 4);
}

// ---- bx matrices (a2) ----
void trh_mtx_srt(float *m, const float *s3, const float *r3, const float *t3) {
    bx::mtxSRT(m, s3[0], s3[1], s3[2], r3[0], r3[1], r3[2], t3[0], t3[1], t3[2]);
}
void trh_mtx_inverse(float *out, const float *in) { bx::mtxInverse(out, in); }

// ---- camera + uniforms without a GPU (a2, a3): a Renderer subclass that only does the host math ----
namespace {
struct CameraProbe : public HipRenderer {
    void setup(int w, int h) { m_width = w; m_height = h; m_aspectRatio = float(w) / float(h); }
    void setFrame(int f) { m_frameIndex = f; }
};
}  // namespace
void trh_uniforms(int w, int h, int frameIndex, const float *eye3, const float *at3, void *uniforms176, float *viewProj16) {
    CameraProbe r;
    r.setup(w, h);
    r.setCameraPosition(bx::Vec3(eye3[0], eye3[1], eye3[2]));
    r.setCameraLookAt(bx::Vec3(at3[0], at3[1], at3[2]));
    r.setFrame(frameIndex);
    Uniforms u;
    r.fillUniforms(&u);
    memcpy(uniforms176, &u, sizeof(u));
    if (viewProj16) r.getViewProjMtx(viewProj16);
}

// ---- random texture (a4) ----
void trh_random_texture(int w, int h, uint32_t seed, uint32_t *out) {
    Texture t = Texture::generateRandomTexture(w, h, 4, seed);
    memcpy(out, t.getBufferPointer(), t.getBufferSize());
    t.destroy();
}

// ---- asynchronous frame loop with a camera move in the middle: framesA x renderFrame(), setCameraPosition(eyeB), framesB x
//      renderFrame(), read back.  Frames queued before the move must be rendered with the old camera. ----
// nDevices > 0: the renderer is put on a device group first (HipRenderer::setDevices; one device is a group of one and goes through
// the same trg_group_* calls, launch fences included).
int trh_async_camera_move_on(int w, int h, int framesA, int framesB, const float *eyeB3, int bounces, int device, const int *devices, int nDevices,
                             float *accumOut, unsigned int *launchesOut) {
    Engine *engine = Engine::instance();
    engine->setDevice(nDevices > 0 ? devices[0] : device);
    engine->init(w, h);
    HipRenderer r;
    if (nDevices > 0 && !r.setDevices(devices, nDevices)) return -2;
    if (!r.init()) return -3;
    if (r.getDeviceCount() != (nDevices > 0 ? nDevices : 0)) return -6;
    r.setCameraPosition(bx::Vec3(0.0f, 1.0f, 3.38f));
    r.setCameraLookAt(bx::Vec3(0.0f, 1.0f, -1.0f));
    Scene *scene = createCornellBoxScene();
    r.loadScene(scene);
    delete scene;
    r.setBounces((unsigned int)bounces);
    for (int i = 0; i < framesA; ++i) r.renderFrame();
    r.setCameraPosition(bx::Vec3(eyeB3[0], eyeB3[1], eyeB3[2]));   // a base-class setter: the backend cannot intercept it
    for (int i = 0; i < framesB; ++i) r.renderFrame();
    if (r.getFrameIndex() != framesA + framesB) return -4;
    if (!r.readAccumulation(accumOut)) return -5;
    if (launchesOut) *launchesOut = r.getLaunchCount();
    return 0;
}
int trh_async_camera_move(int w, int h, int framesA, int framesB, const float *eyeB3, int bounces, int device, float *accumOut, unsigned int *launchesOut) {
    return trh_async_camera_move_on(w, h, framesA, framesB, eyeB3, bounces, device, nullptr, 0, accumOut, launchesOut);
}

// ---- HipRenderer on a caller-built Scene: init, setDeviceBuild, loadScene, `frames` samples, read back ----
int trh_render_scene_on(void *scene, int w, int h, int frames, int bounces, int deviceBuild, int device, const int *devices, int nDevices,
                        float *accumOut, double *loadMsOut) {
    Engine *engine = Engine::instance();
    engine->setDevice(nDevices > 0 ? devices[0] : device);
    engine->init(w, h);
    HipRenderer r;
    if (nDevices > 0 && !r.setDevices(devices, nDevices)) return -2;
    if (!r.init()) return -3;
    if (!r.setDeviceBuild(deviceBuild)) return -2;
    r.setCameraPosition(bx::Vec3(0.0f, 1.0f, 3.38f));
    r.setCameraLookAt(bx::Vec3(0.0f, 1.0f, -1.0f));
    const auto t0 = std::chrono::steady_clock::now();
    r.loadScene(static_cast<Scene *>(scene));
    if (loadMsOut) *loadMsOut = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    r.setBounces((unsigned int)bounces);
    if (!r.renderFrames((unsigned int)frames)) return -4;
    if (!r.readAccumulation(accumOut)) return -5;
    return 0;
}
int trh_render_scene(void *scene, int w, int h, int frames, int bounces, int deviceBuild, int device, float *accumOut, double *loadMsOut) {
    return trh_render_scene_on(scene, w, h, frames, bounces, deviceBuild, device, nullptr, 0, accumOut, loadMsOut);
}

// ---- the reference app's call sequence (main.cpp:21-95), headless ----
// returns 0 on success; negative = the step that failed
int trh_run_app(int w, int h, int frames, int bounces, int batch, int device, float *accumOut, const char *pngPath,
                double *msOut, unsigned long long *raysOut, unsigned int *launchesOut) {
    Engine *engine = Engine::instance();
    engine->setDevice(device);
    engine->init(w, h);
    engine->setFrameBudget(frames);

    const char *names[3] = { "Raytracing", "Accumulate", "PostProcessing" };
    Shader *shaders[3];
    for (int i = 0; i < 3; ++i) {
        shaders[i] = Engine::createShader();
        if (!shaders[i]->load(names[i])) return -1;
    }
    shaders[0]->addFunction("raygen", ShaderFunctionType::RayGen);
    shaders[0]->addFunction("primaryHit", ShaderFunctionType::ClosestHit);
    shaders[0]->addFunction("primaryMiss", ShaderFunctionType::Miss);
    shaders[0]->addFunction("shadowHit", ShaderFunctionType::ShadowHit);
    shaders[0]->addFunction("shadowMiss", ShaderFunctionType::ShadowMiss);
    shaders[1]->addFunction("accumulate", ShaderFunctionType::Compute);
    shaders[2]->addFunction("vert", ShaderFunctionType::Vertex);
    shaders[2]->addFunction("frag", ShaderFunctionType::Fragment);
    if (!shaders[0]->compile(ShaderType::Raytrace) || !shaders[1]->compile(ShaderType::Compute) ||
        !shaders[2]->compile(ShaderType::Graphics))
        return -2;

    Renderer *renderer = Engine::createRenderer();
    if (!renderer->init()) return -3;
    for (int i = 0; i < 3; ++i) renderer->addShader(shaders[i]);
    renderer->setCameraPosition(bx::Vec3(0.0f, 1.0f, 3.38f));
    renderer->setCameraLookAt(bx::Vec3(0.0f, 1.0f, -1.0f));

    Scene *scene = createCornellBoxScene();
    renderer->loadScene(scene);
    delete scene;  // loadScene borrows only for the call

    HipRenderer *hip = static_cast<HipRenderer *>(renderer);
    hip->setBounces((unsigned int)bounces);
    int rc = 0;
    hip->finish();
    const auto t0 = std::chrono::steady_clock::now();
    if (batch) {
        if (!hip->renderFrames((unsigned int)frames)) rc = -4;
    } else {
        while (!engine->hasQuit()) {  // main.cpp:91-95
            engine->pollEvents();
            if (engine->hasQuit()) break;
            renderer->renderFrame();
        }
        if (hip->getFrameIndex() != frames) rc = -4;
    }
    hip->finish();   // the loop above only enqueues: this is where the frames are waited for
    const double loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (launchesOut) *launchesOut = hip->getLaunchCount();
    if (rc == 0 && accumOut && !hip->readAccumulation(accumOut)) rc = -5;
    if (rc == 0 && pngPath && pngPath[0] && !hip->savePNG(pngPath)) rc = -6;
    if (msOut) *msOut = loop_ms;   // wall time of the frame loop, enqueue to completion
    if (raysOut) *raysOut = hip->getRayCount();
    renderer->destroy();
    delete renderer;
    for (int i = 0; i < 3; ++i) delete shaders[i];
    return rc;
}

}  // extern "C"
