// trg_group.cpp -- multi-GPU rendering behind the C ABI (include/trg.h, "trg_group_*"): one process, one context per device,
// one PERSISTENT host thread per context (trg_workers.h), RCCL over xGMI for the one exchange a frame needs.
//
// pixels are independent (every quantity is a function of (x, y, frameIndex); the scene is read-only and
// replicated), so a frame shards by contiguous ROW BANDS: device g renders rows [g*B, min(h, (g+1)*B)), B = ceil(h / G), all its
// samples, into ITS slice of a full-frame accumulation buffer; ONE exchange at the end.  Bands of B rows (the last one may be
// shorter, the buffers are padded to G*B rows) make the in-place ncclAllGather exact for any image height:
// sendbuff = recvbuff + rank * count, the layout NCCL defines as in-place.  TRG_GATHER_ROOT sends the bands to one device only
// (grouped ncclSend / ncclRecv: the root ingests G-1 bands over G-1 independent xGMI links).  No reduction is ever needed.
// The reference has no multi-GPU code (MetalRenderer.mm is single-device); this is new.
//
// RCCL is loaded with dlopen when a group of more than one device is created, so the single-GPU library has no link-time
// dependency on it.
//
// The COPY exchange (round 3; environment variable TRG_GROUP_EXCHANGE=copy, and the fallback when librccl.so cannot be loaded): the same bands
// moved with hipMemcpyPeerAsync on the destination's stream, ordered by events -- a band is copied once its owner has rendered it, and an owner
// does not start its next frame before every copy of its band has been taken.  No communicator, so the devices of such a group need not
// be distinct: a group of n contexts on ONE device runs the whole n > 1 machinery (bands, worker threads, scene uploads, fences, gather)
// on a single-GPU box -- which is what the GPU tests do -- everything but the RCCL calls themselves.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/trg.h"
#include "trg_internal.h"
#include "trg_workers.h"

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) err = std::string("librccl.so lacks ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return CommInitAll && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;
thread_local std::string g_group_create_err;
}  // namespace

struct trg_group {
    int n = 0;
    uint32_t w = 0, h = 0, band = 0;          // band = rows per device = ceil(h / n)
    std::vector<int> devices;
    std::vector<trg_ctx *> ctx;
    std::vector<float *> frame;                // per device: n * band rows x w float4 (padded full frame), bound as its accumulation buffer
    std::vector<ncclComm_t> comm;
    bool copy_exchange = false;                // bands move by hipMemcpyPeerAsync + events instead of RCCL
    std::vector<hipEvent_t> rendered, copied;  // copy exchange: rank r's band is complete / rank r has taken its copies of the others' bands
    std::unique_ptr<trg::DeviceWorkers> workers;   // one persistent host thread per device (none for a group of one)
    std::string err;
};

static int gfail(trg_group *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g) g->err = buf; else g_group_create_err = buf;
    return code;
}

// run f(rank) on every device's own host thread at the same time and collect the first failure
template <typename F>
static int for_each_device(trg_group *g, F f) {
    g->workers->run([&](int r) { return f(r); });
    const int r = g->workers->first_failure();
    if (r >= 0) return gfail(g, g->workers->rc(r), "device %d (rank %d): %s", g->devices[r], r, trg_last_error(g->ctx[r]));
    return TRG_OK;
}

extern "C" {

void trg_band_rows(uint32_t height, uint32_t n, uint32_t r, uint32_t *row0, uint32_t *rows) {
    const uint32_t band = n ? (height + n - 1) / n : height;
    const uint64_t a = (uint64_t)band * r, b = a + band;
    const uint32_t lo = (uint32_t)(a < height ? a : height), hi = (uint32_t)(b < height ? b : height);
    if (row0) *row0 = lo;
    if (rows) *rows = hi - lo;
}

const char *trg_group_last_error(trg_group *g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }
int trg_group_size(trg_group *g) { return g ? g->n : 0; }
int trg_group_exchange(trg_group *g) { return !g ? 0 : (g->copy_exchange ? TRG_EXCHANGE_COPY : (g->comm.empty() ? TRG_EXCHANGE_NONE : TRG_EXCHANGE_RCCL)); }
trg_ctx *trg_group_ctx(trg_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }

void trg_group_destroy(trg_group *g) {
    if (!g) return;
    g->workers.reset();   // joins the host threads (none of them has work: every entry point waits for its jobs)
    for (int r = 0; r < g->n; ++r) {
        if (r < (int)g->ctx.size() && g->ctx[r]) { (void)trg_sync(g->ctx[r]); }
    }
    for (size_t r = 0; r < g->comm.size(); ++r)
        if (g->comm[r]) (void)g_rccl.CommDestroy(g->comm[r]);
    for (size_t r = 0; r < g->rendered.size(); ++r) {
        (void)hipSetDevice(g->devices[r]);
        if (g->rendered[r]) (void)hipEventDestroy(g->rendered[r]);
        if (r < g->copied.size() && g->copied[r]) (void)hipEventDestroy(g->copied[r]);
    }
    for (int r = 0; r < g->n; ++r) {
        if (r < (int)g->ctx.size() && g->ctx[r]) trg_destroy(g->ctx[r]);
        if (r < (int)g->frame.size() && g->frame[r]) { (void)hipSetDevice(g->devices[r]); (void)hipFree(g->frame[r]); }
    }
    delete g;
}

int trg_group_create(trg_group **out, const int *devices, int n, uint32_t width, uint32_t height) {
    if (!out || !devices || n < 1 || n > 64 || width == 0 || height == 0) return gfail(nullptr, TRG_ERR_INVALID, "trg_group_create: bad arguments");
    *out = nullptr;
    const char *xenv = getenv("TRG_GROUP_EXCHANGE");
    if (xenv && strcmp(xenv, "copy") != 0 && strcmp(xenv, "rccl") != 0) return gfail(nullptr, TRG_ERR_INVALID, "trg_group_create: TRG_GROUP_EXCHANGE must be rccl or copy");
    bool copy_exchange = xenv && strcmp(xenv, "copy") == 0;
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (devices[a] == devices[b] && !copy_exchange)   // (an RCCL communicator needs distinct devices; the copy exchange does not)
                return gfail(nullptr, TRG_ERR_INVALID, "trg_group_create: device %d listed twice", devices[a]);
    trg_group *g = new (std::nothrow) trg_group;
    if (!g) return gfail(nullptr, TRG_ERR_NOMEM, "trg_group_create: out of host memory");
    g->n = n; g->w = width; g->h = height; g->band = (height + (uint32_t)n - 1) / (uint32_t)n;
    g->devices.assign(devices, devices + n);
    g->ctx.assign(n, nullptr); g->frame.assign(n, nullptr);
    const size_t frame_bytes = (size_t)g->band * n * width * 16u;
    for (int r = 0; r < n; ++r) {
        int rc = trg_create(&g->ctx[r], devices[r], width, height);
        if (rc != TRG_OK) { gfail(nullptr, rc, "trg_group_create: device %d: %s", devices[r], trg_last_error(nullptr)); trg_group_destroy(g); return rc; }
        hipError_t e = hipSetDevice(devices[r]);
        if (e == hipSuccess) e = hipMalloc((void **)&g->frame[r], frame_bytes);
        if (e == hipSuccess) e = hipMemset(g->frame[r], 0, frame_bytes);
        if (e != hipSuccess) {
            gfail(nullptr, TRG_ERR_NOMEM, "trg_group_create: device %d: %s", devices[r], hipGetErrorString(e));
            trg_group_destroy(g);
            return TRG_ERR_NOMEM;
        }
        (void)trg_bind_accum(g->ctx[r], g->frame[r]);
    }
    // (TRG_GROUP_FORCE_RCCL: a group of ONE device goes through RCCL too -- dlopen, ncclCommInitAll, the in-place all-gather on the render's
    // stream -- so that as much of the exchange as one GPU can exercise is exercised by the GPU tests of a single-GPU box)
    if (!copy_exchange && (n > 1 || getenv("TRG_GROUP_FORCE_RCCL"))) {
        std::string err;
        if (!g_rccl.load(err)) {
            if (getenv("TRG_GROUP_FORCE_RCCL") || xenv) { gfail(nullptr, TRG_ERR_DEVICE, "trg_group_create: %s", err.c_str()); trg_group_destroy(g); return TRG_ERR_DEVICE; }
            copy_exchange = true;   // no librccl.so on this machine: the bands move by peer copies
        }
    }
    if (!copy_exchange && (n > 1 || getenv("TRG_GROUP_FORCE_RCCL"))) {
        g->comm.assign(n, nullptr);
        const ncclResult_t nr = g_rccl.CommInitAll(g->comm.data(), n, g->devices.data());
        if (nr != ncclSuccess) {
            gfail(nullptr, TRG_ERR_DEVICE, "trg_group_create: ncclCommInitAll: %s", g_rccl.GetErrorString(nr));
            g->comm.clear();
            trg_group_destroy(g);
            return TRG_ERR_DEVICE;
        }
    }
    g->copy_exchange = copy_exchange && n > 1;
    if (g->copy_exchange) {
        g->rendered.assign(n, nullptr); g->copied.assign(n, nullptr);
        for (int r = 0; r < n; ++r) {
            hipError_t e = hipSetDevice(devices[r]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&g->rendered[r], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&g->copied[r], hipEventDisableTiming);
            if (e != hipSuccess) { gfail(nullptr, TRG_ERR_DEVICE, "trg_group_create: device %d: %s", devices[r], hipGetErrorString(e)); trg_group_destroy(g); return TRG_ERR_DEVICE; }
            for (int p = 0; p < n; ++p) {   // direct copies over xGMI where the devices can reach each other (otherwise the runtime stages them)
                int can = 0;
                if (devices[p] != devices[r] && hipDeviceCanAccessPeer(&can, devices[r], devices[p]) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(devices[p], 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
        }
        (void)hipGetLastError();
    }
    // the devices' host threads: started once, each bound to its device; a frame is handed to them through a condition variable
    g->workers.reset(new (std::nothrow) trg::DeviceWorkers(n, [g](int r) { (void)hipSetDevice(g->devices[r]); }));
    if (!g->workers) { gfail(nullptr, TRG_ERR_NOMEM, "trg_group_create: out of host memory"); trg_group_destroy(g); return TRG_ERR_NOMEM; }
    *out = g;
    return TRG_OK;
}

int trg_group_load_scene(trg_group *g, const float *positions3, const float *normals3, const float *colors3, const uint32_t *indices,
                         const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris) {
    if (!g) return TRG_ERR_INVALID;
    // a device build (TRG_OPT_GPU_BUILD) runs on every device, at the same time; the host build runs ONCE here and its blob is
    // uploaded to every device in parallel (round 2 ran the same SAH build on all N threads)
    if (trg::ctx_gpu_build_option(g->ctx[0]) && n_tris >= 2)
        return for_each_device(g, [&](int r) { return trg_load_scene(g->ctx[r], positions3, normals3, colors3, indices, material_ids, n_verts, n_tris); });
    trg::HostScene *hs = nullptr;
    if (int rc = trg::host_scene_build(g->ctx[0], positions3, normals3, colors3, indices, material_ids, n_verts, n_tris, &hs))
        return gfail(g, rc, "%s", trg_last_error(g->ctx[0]));
    const int rc = for_each_device(g, [&](int r) { return trg::host_scene_upload(g->ctx[r], hs); });
    trg::host_scene_free(hs);
    return rc;
}
int trg_group_load_textures(trg_group *g, const float *uv2, const uint32_t *texture_ids, uint32_t n_tris, const uint8_t *const *images_rgba8,
                            const uint32_t *widths, const uint32_t *heights, uint32_t n_textures) {
    if (!g) return TRG_ERR_INVALID;
    return for_each_device(g, [&](int r) { return trg_load_textures(g->ctx[r], uv2, texture_ids, n_tris, images_rgba8, widths, heights, n_textures); });
}
int trg_group_set_uniforms(trg_group *g, const trg_uniforms *u) {
    if (!g) return TRG_ERR_INVALID;
    for (int r = 0; r < g->n; ++r)
        if (int rc = trg_set_uniforms(g->ctx[r], u)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
    return TRG_OK;
}
int trg_group_set_pixel_offsets_seed(trg_group *g, uint32_t seed) {
    if (!g) return TRG_ERR_INVALID;
    return for_each_device(g, [&](int r) { return trg_set_pixel_offsets_seed(g->ctx[r], seed); });
}
int trg_group_set_option(trg_group *g, int option, int64_t value) {
    if (!g) return TRG_ERR_INVALID;
    for (int r = 0; r < g->n; ++r)
        if (int rc = trg_set_option(g->ctx[r], option, value)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
    return TRG_OK;
}

int trg_group_render(trg_group *g, uint32_t frameIndexBegin, uint32_t spp, uint32_t bounces, int gather, int root) {
    if (!g) return TRG_ERR_INVALID;
    if (gather != TRG_GATHER_NONE && gather != TRG_GATHER_ALL && gather != TRG_GATHER_ROOT) return gfail(g, TRG_ERR_INVALID, "trg_group_render: unknown gather mode %d", gather);
    if (gather == TRG_GATHER_ROOT && (root < 0 || root >= g->n)) return gfail(g, TRG_ERR_INVALID, "trg_group_render: root %d out of range", root);
    // every device renders its band (one host thread per context)
    if (int rc = for_each_device(g, [&](int r) {
            uint32_t row0, rows;
            trg_band_rows(g->h, (uint32_t)g->n, (uint32_t)r, &row0, &rows);
            return trg_render(g->ctx[r], frameIndexBegin, spp, bounces, row0, rows);
        }))
        return rc;
    if (gather == TRG_GATHER_NONE) return TRG_OK;
    if (g->copy_exchange) {
        // the same exchange by copies: destination d takes band p from its owner once p has rendered it (event), on d's stream; then no
        // owner goes on (its next frame accumulates into its band in place) before every destination has taken its copy
        const size_t count = (size_t)g->band * g->w * 4u, bytes = count * sizeof(float);
        int caller_device = -1;
        (void)hipGetDevice(&caller_device);   // (the calls below select devices on the CALLER's thread: put its device back afterwards)
        hipError_t e = hipSuccess;
        for (int r = 0; r < g->n && e == hipSuccess; ++r) {
            e = hipSetDevice(g->devices[r]);
            if (e == hipSuccess) e = hipEventRecord(g->rendered[r], static_cast<hipStream_t>(trg::ctx_current_stream(g->ctx[r])));
        }
        for (int d = 0; d < g->n && e == hipSuccess; ++d) {
            if (gather == TRG_GATHER_ROOT && d != root) continue;
            hipStream_t s = static_cast<hipStream_t>(trg::ctx_current_stream(g->ctx[d]));
            e = hipSetDevice(g->devices[d]);
            for (int p = 0; p < g->n && e == hipSuccess; ++p) {
                if (p == d) continue;
                e = hipStreamWaitEvent(s, g->rendered[p], 0);
                if (e == hipSuccess)
                    e = g->devices[d] == g->devices[p]
                            ? hipMemcpyAsync(g->frame[d] + (size_t)p * count, g->frame[p] + (size_t)p * count, bytes, hipMemcpyDeviceToDevice, s)
                            : hipMemcpyPeerAsync(g->frame[d] + (size_t)p * count, g->devices[d], g->frame[p] + (size_t)p * count, g->devices[p], bytes, s);
            }
            if (e == hipSuccess) e = hipEventRecord(g->copied[d], s);
        }
        for (int p = 0; p < g->n && e == hipSuccess; ++p) {
            e = hipSetDevice(g->devices[p]);
            for (int d = 0; d < g->n && e == hipSuccess; ++d) {
                if (d == p || (gather == TRG_GATHER_ROOT && d != root)) continue;
                e = hipStreamWaitEvent(static_cast<hipStream_t>(trg::ctx_current_stream(g->ctx[p])), g->copied[d], 0);
            }
        }
        if (caller_device >= 0) (void)hipSetDevice(caller_device);
        if (e != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_render: copy exchange: %s", hipGetErrorString(e));
        return TRG_OK;
    }
    if (g->comm.empty()) return TRG_OK;
    // the one exchange of the frame, enqueued behind each device's render: on the stream that render was launched on (the context's
    // current stream -- trg_set_stream's if the caller set one through trg_group_ctx, the context's own otherwise)
    const size_t count = (size_t)g->band * g->w * 4u;   // floats per band
    ncclResult_t nr = g_rccl.GroupStart();
    for (int r = 0; r < g->n && nr == ncclSuccess; ++r) {
        hipStream_t s = static_cast<hipStream_t>(trg::ctx_current_stream(g->ctx[r]));
        float *mine = g->frame[r] + (size_t)r * count;
        if (gather == TRG_GATHER_ALL) {
            nr = g_rccl.AllGather(mine, g->frame[r], count, ncclFloat, g->comm[r], s);   // in place: sendbuff = recvbuff + rank * count
        } else if (r == root) {
            for (int p = 0; p < g->n && nr == ncclSuccess; ++p)
                if (p != root) nr = g_rccl.Recv(g->frame[r] + (size_t)p * count, count, ncclFloat, p, g->comm[r], s);
        } else {
            nr = g_rccl.Send(mine, count, ncclFloat, root, g->comm[r], s);
        }
    }
    const ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_render: RCCL: %s", g_rccl.GetErrorString(nr));
    return TRG_OK;
}

int trg_group_sync(trg_group *g) {
    if (!g) return TRG_ERR_INVALID;
    return for_each_device(g, [&](int r) { return trg_sync(g->ctx[r]); });
}

// frames in flight (MetalRenderer.mm:33,377,385-387: a semaphore of three): a mark on EVERY device's stream, and the wait for all of them
int trg_group_fence_record(trg_group *g, int slot) {
    if (!g) return TRG_ERR_INVALID;
    for (int r = 0; r < g->n; ++r)
        if (int rc = trg_fence_record(g->ctx[r], slot)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
    return TRG_OK;
}
int trg_group_fence_wait(trg_group *g, int slot) {
    if (!g) return TRG_ERR_INVALID;
    for (int r = 0; r < g->n; ++r)
        if (int rc = trg_fence_wait(g->ctx[r], slot)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
    return TRG_OK;
}

int trg_group_read_accum(trg_group *g, int rank, float *rgba) {
    if (!g || rank < 0 || rank >= g->n || !rgba) return TRG_ERR_INVALID;
    if (int rc = trg_read_accum(g->ctx[rank], rgba)) return gfail(g, rc, "rank %d: %s", rank, trg_last_error(g->ctx[rank]));
    return TRG_OK;
}

int trg_group_get_stats(trg_group *g, trg_stats *out) {
    if (!g || !out) return TRG_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    for (int r = 0; r < g->n; ++r) {
        trg_stats st;
        if (int rc = trg_get_stats(g->ctx[r], &st)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
        if (r == 0) *out = st;
        else {
            out->primary_rays += st.primary_rays; out->bounce_rays += st.bounce_rays; out->shadow_rays += st.shadow_rays; out->shaded_hits += st.shaded_hits;
            out->node_fetches += st.node_fetches; out->tri_tests += st.tri_tests; out->wave_node_iters += st.wave_node_iters; out->wave_tri_iters += st.wave_tri_iters;
            if (st.last_render_ms > out->last_render_ms) out->last_render_ms = st.last_render_ms;   // the slowest device
            if (st.total_render_ms > out->total_render_ms) out->total_render_ms = st.total_render_ms;
        }
    }
    return TRG_OK;
}

}  // extern "C"
