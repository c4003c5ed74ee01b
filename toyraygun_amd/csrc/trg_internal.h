// trg_internal.h -- what trg_group.cpp needs from trg_capi.cpp beyond the public C ABI (hidden visibility: not exported).
#pragma once
#include <stdint.h>

#include "../../include/trg.h"

namespace trg {

// A scene built ON THE HOST -- acceleration structure + the image of the device blob -- ready to be uploaded to any number of
// contexts: a device group builds once and uploads N times instead of running the same build on every device's thread.
struct HostScene;
// builds (validates the buffers like trg_load_scene; errors go to `c`).  *out is owned by the caller: host_scene_free.
int host_scene_build(trg_ctx *c, const float *positions3, const float *normals3, const float *colors3, const uint32_t *indices,
                     const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris, HostScene **out);
int host_scene_upload(trg_ctx *c, const HostScene *hs);
void host_scene_free(HostScene *hs);

int ctx_gpu_build_option(const trg_ctx *c);   // TRG_OPT_GPU_BUILD of the context (a device build runs per device)
void *ctx_current_stream(trg_ctx *c);         // the stream trg_render launches on: trg_set_stream's, or the context's own
int ctx_device(const trg_ctx *c);
void ctx_forget_streams(trg_ctx *c);          // a DRAINED context forgets which streams owned its per-stream scratch slots (their buffers stay)

}  // namespace trg
