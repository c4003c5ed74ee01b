// HipRenderer.cpp -- toyraygun::Renderer backend for MI355X (see include/engine/HipRenderer.h).
// Everything that touches the GPU goes through the C ABI of include/trg.h.
#include "engine/HipRenderer.h"

#include <stdio.h>
#include <string.h>

#include <vector>

#include "png_writer.h"
#include "trg.h"

namespace toyraygun {

HipRenderer::HipRenderer()
    : m_ctx(nullptr), m_group(nullptr), m_deviceCount(0), m_bounces(3), m_deviceBuild(0), m_offsetSeed(0x5EED0001u), m_sceneLoaded(false), m_synchronous(false), m_pending(0), m_pendingFirst(0),
      m_launches(0) {
    memset(&m_pendingUniforms, 0, sizeof(m_pendingUniforms));
}
HipRenderer::~HipRenderer() { destroy(); }

bool HipRenderer::setDevices(const int *devices, int count) {
    if (m_ctx || !devices || count < 1 || count > 16) return false;   // before init() only
    for (int i = 0; i < count; ++i) m_devices[i] = devices[i];
    m_deviceCount = count;
    return true;
}

bool HipRenderer::init() {
    Renderer::init();  // width/height/aspect from the Engine (Renderer.cpp:18-27)
    destroy();
    if (m_deviceCount >= 1) {   // setDevices() was called: a device group, also for one device (same calls, nothing to exchange)
        const int rc = trg_group_create(&m_group, m_devices, m_deviceCount, (uint32_t)m_width, (uint32_t)m_height);
        if (rc != TRG_OK) {
            printf("HipRenderer: %s\n", trg_group_last_error(nullptr));
            m_group = nullptr;
            return false;
        }
        m_ctx = trg_group_ctx(m_group, 0);
        if (trg_group_set_pixel_offsets_seed(m_group, m_offsetSeed) != TRG_OK) {
            printf("HipRenderer: %s\n", trg_group_last_error(m_group));
            return false;
        }
    } else {
        const int dev = Engine::instance()->getDevice();
        const int rc = trg_create(&m_ctx, dev, (uint32_t)m_width, (uint32_t)m_height);
        if (rc != TRG_OK) {
            printf("HipRenderer: %s\n", trg_last_error(nullptr));
            m_ctx = nullptr;
            return false;
        }
        // the reference creates the random-offset texture in resize() (MetalRenderer.mm:323-335)
        if (trg_set_pixel_offsets_seed(m_ctx, m_offsetSeed) != TRG_OK) {
            printf("HipRenderer: %s\n", trg_last_error(m_ctx));
            return false;
        }
    }
    m_frameIndex = 0;  // MetalRenderer.mm:337
    m_sceneLoaded = false;
    m_pending = 0; m_launches = 0;
    // asynchronous launches: nothing waits per frame
    if (m_group) trg_group_set_option(m_group, TRG_OPT_TIMING, m_synchronous ? 1 : 0);
    else trg_set_option(m_ctx, TRG_OPT_TIMING, m_synchronous ? 1 : 0);
    return true;
}

void HipRenderer::destroy() {
    if (m_ctx) flush();
    if (m_group) trg_group_destroy(m_group);   // owns every context, m_ctx included
    else if (m_ctx) trg_destroy(m_ctx);
    m_group = nullptr;
    m_ctx = nullptr;
    m_sceneLoaded = false;
}

void HipRenderer::loadScene(Scene *scene) {
    if (!m_ctx || !scene) return;
    flush();   // frames queued for the old scene are rendered with it
    static_assert(sizeof(bx::Vec3) == 12, "bx::Vec3 must be 3 packed floats");
    const uint32_t nTris = (uint32_t)scene->m_materialIDBuffer.size();
    const uint32_t nVerts = (uint32_t)scene->m_vertexBuffer.size();
    const float *pos = nVerts ? &scene->m_vertexBuffer[0].x : nullptr;
    const float *nrm = nVerts ? &scene->m_normalBuffer[0].x : nullptr;
    const float *col = nVerts ? &scene->m_colorBuffer[0].x : nullptr;
    const uint32_t *idx = nVerts ? &scene->m_indexBuffer[0] : nullptr;
    const uint32_t *mat = nTris ? &scene->m_materialIDBuffer[0] : nullptr;
    // a device-built tree is traversed in HBM: scenes of a few hundred triangles are staged in LDS and keep the host builder
    const int builder = nTris >= 1024u ? m_deviceBuild : 0;
    if (m_group) trg_group_set_option(m_group, TRG_OPT_GPU_BUILD, builder);
    else trg_set_option(m_ctx, TRG_OPT_GPU_BUILD, builder);
    const int rc = m_group ? trg_group_load_scene(m_group, pos, nrm, col, idx, mat, nVerts, nTris)
                           : trg_load_scene(m_ctx, pos, nrm, col, idx, mat, nVerts, nTris);
    if (rc != TRG_OK) {
        printf("HipRenderer: %s\n", m_group ? trg_group_last_error(m_group) : trg_last_error(m_ctx));
        m_sceneLoaded = false;
        return;
    }
    m_sceneLoaded = true;
    // albedo textures of the scene (Scene::addMesh with a Texture): expanded to RGBA8 and uploaded with the texture coordinates
    if (!scene->m_textures.empty() && nTris) {
        std::vector<float> uv(scene->m_uvBuffer);
        std::vector<uint32_t> ids(scene->m_textureIDBuffer);
        uv.resize((size_t)nVerts * 2, 0.0f);       // triangles added after the last textured mesh
        ids.resize(nTris, 0u);
        std::vector<std::vector<uint8_t> > rgba(scene->m_textures.size());
        std::vector<const uint8_t *> ptrs;
        std::vector<uint32_t> ws, hs;
        for (size_t k = 0; k < scene->m_textures.size(); ++k) {
            Texture *t = scene->m_textures[k];
            const int w = t->getWidth(), h = t->getHeight(), ch = t->getChannels();
            const uint8_t *src = t->getBufferPointer();
            rgba[k].resize((size_t)w * h * 4);
            for (size_t i = 0; i < (size_t)w * h; ++i) {
                const uint8_t *p = src + i * ch;
                uint8_t *q = &rgba[k][i * 4];
                if (ch >= 3) { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = ch == 4 ? p[3] : 255; }
                else { q[0] = q[1] = q[2] = p[0]; q[3] = ch == 2 ? p[1] : 255; }
            }
            ptrs.push_back(rgba[k].data()); ws.push_back((uint32_t)w); hs.push_back((uint32_t)h);
        }
        const int rc = m_group ? trg_group_load_textures(m_group, uv.data(), ids.data(), nTris, ptrs.data(), ws.data(), hs.data(), (uint32_t)ptrs.size())
                               : trg_load_textures(m_ctx, uv.data(), ids.data(), nTris, ptrs.data(), ws.data(), hs.data(), (uint32_t)ptrs.size());
        if (rc != TRG_OK) {
            printf("HipRenderer: %s\n", m_group ? trg_group_last_error(m_group) : trg_last_error(m_ctx));
            m_sceneLoaded = false;
            return;
        }
    }
}

// MetalRenderer.mm:340-371: inverse view-projection, fixed ceiling light, frame index.
void HipRenderer::fillUniforms(Uniforms *u) {
    memset(u, 0, sizeof(*u));
    float viewProj[16], invViewProj[16];
    getViewProjMtx(viewProj);
    bx::mtxInverse(invViewProj, viewProj);
    u->camera.position.set(getCameraPosition());
    u->camera.invViewProjMtx.set(invViewProj);
    u->light.position.set(bx::Vec3(0.0f, 1.98f, 0.0f));
    u->light.forward.set(bx::Vec3(0.0f, -1.0f, 0.0f));
    u->light.right.set(bx::Vec3(0.25f, 0.0f, 0.0f));
    u->light.up.set(bx::Vec3(0.0f, 0.0f, 0.25f));
    u->light.color.set(bx::Vec3(1.0f, 1.0f, 1.0f));
    u->width = (unsigned int)m_width;
    u->height = (unsigned int)m_height;
    u->frameIndex = (unsigned int)m_frameIndex;
}

// Launch the queued frames [m_pendingFirst, m_pendingFirst + m_pending) in one megakernel launch.  A semaphore of
// kFramesInFlight launches (MetalRenderer.mm:377: dispatch_semaphore_wait before encoding) bounds what is queued on the device.
bool HipRenderer::flush() {
    if (!m_ctx || m_pending == 0) return true;
    const unsigned int frames = m_pending;
    m_pending = 0;
    if (!m_sceneLoaded) return false;
    static_assert(sizeof(Uniforms) == sizeof(trg_uniforms), "Uniforms / trg_uniforms layout mismatch");
    const int slot = (int)(m_launches % (unsigned int)kFramesInFlight);
    if (m_group) {   // row bands over the devices, gathered on the first one (which presents); the same bound on launches in flight
        if (trg_group_fence_wait(m_group, slot) != TRG_OK ||
            trg_group_set_uniforms(m_group, reinterpret_cast<const trg_uniforms *>(&m_pendingUniforms)) != TRG_OK ||
            trg_group_render(m_group, (uint32_t)m_pendingFirst, frames, m_bounces, TRG_GATHER_ROOT, 0) != TRG_OK ||
            trg_group_fence_record(m_group, slot) != TRG_OK) {
            printf("HipRenderer: %s\n", trg_group_last_error(m_group));
            return false;
        }
        ++m_launches;
        return true;
    }
    if (trg_fence_wait(m_ctx, slot) != TRG_OK ||
        trg_set_uniforms(m_ctx, reinterpret_cast<const trg_uniforms *>(&m_pendingUniforms)) != TRG_OK ||
        trg_render(m_ctx, (uint32_t)m_pendingFirst, frames, m_bounces, 0, (uint32_t)m_height) != TRG_OK ||
        trg_fence_record(m_ctx, slot) != TRG_OK) {
        printf("HipRenderer: %s\n", trg_last_error(m_ctx));
        return false;
    }
    ++m_launches;
    return true;
}

// Accept `frames` more frames with the CURRENT uniforms.  Frames whose uniforms differ from the queued ones cannot share a launch.
static bool sameUniforms(const Uniforms &a, const Uniforms &b) { return memcmp(&a, &b, sizeof(Uniforms)) == 0; }
bool HipRenderer::renderFrames(unsigned int frames) {
    if (!m_ctx || !m_sceneLoaded || frames == 0) return false;
    Uniforms u;
    fillUniforms(&u);
    u.frameIndex = 0;   // trg_render iterates its own frame range (uniforms->frameIndex = _frameIndex++, MetalRenderer.mm:363)
    if (m_pending && !sameUniforms(u, m_pendingUniforms) && !flush()) return false;
    if (m_pending == 0) { m_pendingUniforms = u; m_pendingFirst = m_frameIndex; }
    m_pending += frames;
    m_frameIndex += (int)frames;
    // launch now if the device has nothing to do (or we are asked to be synchronous, or a lot has piled up); otherwise the
    // frames wait for the next call and share its launch
    if (m_synchronous || frames > 1 || m_pending >= 64u || trg_stream_idle(m_ctx) != 0) return flush();
    return true;
}

void HipRenderer::renderFrame() { renderFrames(1); }

void HipRenderer::setSynchronous(bool on) {
    flush();
    m_synchronous = on;
    if (m_group) trg_group_set_option(m_group, TRG_OPT_TIMING, on ? 1 : 0);
    else if (m_ctx) trg_set_option(m_ctx, TRG_OPT_TIMING, on ? 1 : 0);
}

void HipRenderer::setBounces(unsigned int bounces) { flush(); m_bounces = bounces; }
bool HipRenderer::setDeviceBuild(int builder) {
    if (builder < 0 || builder > 3) return false;
    m_deviceBuild = builder;
    return true;
}
bool HipRenderer::finish() { return m_ctx && flush() && (m_group ? trg_group_sync(m_group) : trg_sync(m_ctx)) == TRG_OK; }
void HipRenderer::setOffsetSeed(uint32_t seed) {
    flush();
    m_offsetSeed = seed;
    if (m_group) trg_group_set_pixel_offsets_seed(m_group, seed);
    else if (m_ctx) trg_set_pixel_offsets_seed(m_ctx, seed);
}
bool HipRenderer::readAccumulation(float *rgbaOut) {
    if (!m_ctx || !flush()) return false;
    if (m_group) return trg_group_read_accum(m_group, 0, rgbaOut) == TRG_OK;   // waits for the gather onto device 0 (and the unpack of interleaved bands)
    return trg_read_accum(m_ctx, rgbaOut) == TRG_OK;
}
bool HipRenderer::savePNG(const char *path) {
    if (!m_ctx || !flush()) return false;
    std::vector<uint8_t> rgba((size_t)m_width * m_height * 4);
    if ((m_group ? trg_group_postprocess(m_group, 0, rgba.data(), 1) : trg_postprocess(m_ctx, rgba.data(), 1)) != TRG_OK) return false;
    return trg_host::write_png_rgba8(path, rgba.data(), m_width, m_height);
}
double HipRenderer::getLastRenderMs() const {
    trg_stats st;
    return (m_ctx && trg_get_stats(m_ctx, &st) == TRG_OK) ? st.last_render_ms : 0.0;
}
uint64_t HipRenderer::getRayCount() const {
    const_cast<HipRenderer *>(this)->flush();
    trg_stats st;
    if (m_group) return trg_group_get_stats(m_group, &st) == TRG_OK ? st.primary_rays + st.bounce_rays + st.shadow_rays : 0;
    return (m_ctx && trg_get_stats(m_ctx, &st) == TRG_OK) ? st.primary_rays + st.bounce_rays + st.shadow_rays : 0;
}
const char *HipRenderer::getLastError() const { return trg_last_error(m_ctx); }

}  // namespace toyraygun
