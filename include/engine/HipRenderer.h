// engine/HipRenderer.h -- the MI355X backend: a toyraygun::Renderer whose loadScene/renderFrame drive the
// gfx950 megakernel through the C ABI of include/trg.h.  Takes the place of MetalRenderer
// (reference src/engine/Metal/MetalRenderer.h:16-29, MetalRenderer.mm:557-597) / D3D12Renderer.
#pragma once
#include <stdint.h>

#include "engine/Renderer.h"

struct trg_ctx;

namespace toyraygun {

class HipRenderer : public Renderer {
public:
    HipRenderer();
    ~HipRenderer() override;

    bool init() override;                   // creates the device context; false (+message on stdout) on failure
    void destroy() override;
    void loadScene(Scene *scene) override;  // uploads the five Scene vectors, builds the BVH
    void renderFrame() override;            // one more sample per pixel into the running average

    // ---- beyond the reference: batch rendering and read-back for headless use ----
    void setBounces(unsigned int bounces);              // reference hard-codes 3 (MetalRenderer.mm:426)
    unsigned int getBounces() const { return m_bounces; }
    void setOffsetSeed(uint32_t seed);                  // per-pixel Halton offset seed (default 0x5EED0001)
    bool renderFrames(unsigned int frames);             // `frames` samples in ONE kernel launch
    bool readAccumulation(float *rgbaOut);              // width*height*4 floats, row 0 = scene bottom
    bool savePNG(const char *path);                     // ACES + sRGB (PostProcessing.metal:44-57), top row first
    int getFrameIndex() const { return m_frameIndex; }
    double getLastRenderMs() const;
    uint64_t getRayCount() const;                       // primary + bounce + shadow rays traced so far
    const char *getLastError() const;
    trg_ctx *context() { return m_ctx; }
    void fillUniforms(Uniforms *out);                   // MetalRenderer.mm:340-371 updateUniforms

protected:
    trg_ctx *m_ctx;
    unsigned int m_bounces;
    uint32_t m_offsetSeed;
    bool m_sceneLoaded;
};

}  // namespace toyraygun
