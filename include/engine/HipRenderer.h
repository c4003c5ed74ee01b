// engine/HipRenderer.h -- the MI355X backend: a toyraygun::Renderer whose loadScene/renderFrame drive the
// gfx950 megakernel through the C ABI of include/trg.h.  Takes the place of MetalRenderer
// (reference src/engine/Metal/MetalRenderer.h:16-29, MetalRenderer.mm:557-597) / D3D12Renderer.
#pragma once
#include <stdint.h>

#include "engine/Renderer.h"

struct trg_ctx;
struct trg_group;

namespace toyraygun {

class HipRenderer : public Renderer {
public:
    HipRenderer();
    ~HipRenderer() override;

    bool init() override;                   // creates the device context; false (+message on stdout) on failure
    void destroy() override;
    void loadScene(Scene *scene) override;  // uploads the five Scene vectors, builds the BVH
    // One more sample per pixel into the running average.  ASYNCHRONOUS like the reference's (MetalRenderer.mm:373-552: encode,
    // commit, return; three frames in flight behind a semaphore, :33,377,385-387): the call never waits for the GPU to finish
    // a frame.  It takes a snapshot of the uniforms and either launches at once (GPU idle) or -- while earlier launches are
    // still running -- lets consecutive frames with identical uniforms ride in ONE later launch (the megakernel renders any
    // number of frames per launch, bit for bit what one launch per frame gives).  At most kFramesInFlight launches are queued
    // on the device; only readAccumulation / savePNG / getRayCount / destroy wait for it.
    void renderFrame() override;

    // ---- beyond the reference: several GPUs of one node.  Call before init(): the frame is then sharded by row bands over the
    //      devices (trg_group_*, include/trg.h) and gathered on the first one, which presents / reads back / writes the PNG.
    //      One device is a group of one (the same calls, nothing to exchange).
    bool setDevices(const int *devices, int count);
    int getDeviceCount() const { return m_deviceCount; }

    // ---- the acceleration structure: the reference rebuilds it on the GPU at loadScene (MPSTriangleAccelerationStructure rebuild,
    //      MetalRenderer.mm:272-279).  0 (default): the host's binned-SAH builder; 1: the same split rule on the device, level by
    //      level (a million triangles in ~20 ms); 2: Karras LBVH; 3: PLOC (TRG_OPT_GPU_BUILD, include/trg.h).  Scenes small enough
    //      to be staged in LDS are always built on the host.  Takes effect at the next loadScene().
    bool setDeviceBuild(int builder);
    int getDeviceBuild() const { return m_deviceBuild; }

    // ---- beyond the reference: batch rendering and read-back for headless use ----
    void setBounces(unsigned int bounces);              // reference hard-codes 3 (MetalRenderer.mm:426)
    unsigned int getBounces() const { return m_bounces; }
    void setOffsetSeed(uint32_t seed);                  // per-pixel Halton offset seed (default 0x5EED0001)
    bool renderFrames(unsigned int frames);             // `frames` samples in ONE kernel launch (asynchronous too)
    bool flush();                                       // launch whatever renderFrame() has queued; does not wait
    bool finish();                                      // flush, then wait until the device has finished every frame
    void setSynchronous(bool on);                       // true: every launch is timed with HIP events and waited for (getLastRenderMs)
    unsigned int getLaunchCount() const { return m_launches; }   // kernel launches so far (progressive loops coalesce frames)
    static const int kFramesInFlight = 3;               // MetalRenderer.mm:33 kMaxFramesInFlight
    bool readAccumulation(float *rgbaOut);              // width*height*4 floats, row 0 = scene bottom
    bool savePNG(const char *path);                     // ACES + sRGB (PostProcessing.metal:44-57), top row first
    int getFrameIndex() const { return m_frameIndex; }
    double getLastRenderMs() const;
    uint64_t getRayCount() const;                       // primary + bounce + shadow rays traced so far
    const char *getLastError() const;
    trg_ctx *context() { return m_ctx; }
    void fillUniforms(Uniforms *out);                   // MetalRenderer.mm:340-371 updateUniforms

protected:
    trg_ctx *m_ctx;              // the context (of the first device when a group is used)
    trg_group *m_group;          // non-null when rendering on several devices
    int m_devices[16];
    int m_deviceCount;
    unsigned int m_bounces;
    int m_deviceBuild;
    uint32_t m_offsetSeed;
    bool m_sceneLoaded;
    bool m_synchronous;
    unsigned int m_pending;      // frames renderFrame() has accepted but not launched yet
    int m_pendingFirst;          // frame index of the first of them
    Uniforms m_pendingUniforms;  // their uniforms (frameIndex zeroed)
    unsigned int m_launches;
};

}  // namespace toyraygun
