// experiments/trg_exp.h -- host-visible interface of the EXPERIMENTAL schedules (the path pool and the wavefront schedule): included by
// toyraygun_amd/csrc/trg_kernels.h when TRG_EXPERIMENTS=1, i.e. only in experiments/lib/libtoyraygun_hip_exp.so (experiments/build.py).
#pragma once

namespace trg {

#ifndef TRG_POOL_S
#define TRG_POOL_S 2
#endif
constexpr int kPoolS = TRG_POOL_S;           // frames (path slots per thread) a pool workgroup keeps in flight
constexpr uint32_t kPoolSlotBytes = 64u;      // R0, R1, SH, H: four float4 per slot

// ---- wavefront schedule (TRG_KERNEL_WAVEFRONT, trg_wavefront.inc.h): path state and ray queues of one batch in HBM ----
constexpr uint32_t kWfMaxPaths = 8u << 20;   // pixel-samples per batch (112 B of state each)
constexpr uint32_t kWfMaxStages = TRG_MAX_BOUNCES + 2u;
constexpr uint32_t kWfPathBytes = 7u * 16u;  // ray_o, ray_d, sh, hit, thr, rad, scol
struct WfBuffers {
    void *ray_o, *ray_d, *sh, *hit, *thr, *rad, *scol;   // float4[nb] each
    uint32_t *list[2];                                   // 2 * nb entries each
    uint32_t *ctr;                                       // 4 per stage: head, count, -, -
};
struct WfParams {
    trg_uniforms u;
    SceneDesc sc;
    const uint32_t *offsets;
    float *accum;
    unsigned long long *counters;
    WfBuffers b;
    TexDesc tex;
    uint32_t nb, npix, pix0, frame0, nframes, bounces, stage, bounce;
    uint32_t stack_off;
    StackDesc stack;
};


#define TRG_DECL_EXP_LAUNCHERS(SFX)                                                                               \
    hipError_t launch_render_pool_##SFX(const RenderParams &p, bool lds_scene, bool counters, uint32_t grid,     \
                                        size_t lds_bytes, hipStream_t s);                                        \
    hipError_t launch_render_rtail_##SFX(const RenderParams &p, bool counters, uint32_t grid, size_t lds_bytes, hipStream_t s); \
    hipError_t launch_wf_raygen_##SFX(const WfParams &p, hipStream_t s);                                         \
    hipError_t launch_wf_trace_##SFX(const WfParams &p, bool lds_scene, bool counters, uint32_t grid, size_t lds_bytes, \
                                     hipStream_t s);                                                             \
    hipError_t launch_wf_shade_##SFX(const WfParams &p, bool lds_scene, uint32_t grid, hipStream_t s);           \
    hipError_t launch_wf_accumulate_##SFX(const WfParams &p, hipStream_t s);

TRG_DECL_EXP_LAUNCHERS(fast)
TRG_DECL_EXP_LAUNCHERS(strict)

}  // namespace trg
