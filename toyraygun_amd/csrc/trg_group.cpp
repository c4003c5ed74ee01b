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
// The COPY exchange (round 3; environment variable TRG_GROUP_EXCHANGE=copy, and the fallback when librccl.so cannot be loaded -- the fallback
// is REPORTED: trg_group_exchange() says TRG_EXCHANGE_COPY and trg_group_last_error() keeps why): the same bands moved with
// hipMemcpyPeerAsync on the destination's exchange stream, ordered by events -- a band is copied once its owner has rendered it, and an
// owner does not render into a frame slot again before every copy of its band has been taken.  Every destination's copies are issued
// from ITS OWN worker thread (round 4; round 3 issued all n * (n - 1) of them from the caller's).  No communicator, so the devices of such
// a group need not be distinct: a group of n contexts on ONE device runs the whole n > 1 machinery (bands, worker threads, scene uploads,
// fences, gather, the SAME hipMemcpyPeerAsync call) on a single-GPU box -- which is what the GPU tests do -- everything but the RCCL
// calls themselves.
//
// PIPELINE (round 4; trg_group_set_pipeline): consecutive trg_group_render calls produce INDEPENDENT images (a benchmark's steps, an
// animation's frames), so a group may keep `depth` of them in flight per device: every device gets `depth` frame buffers, `depth` render
// streams and one exchange stream; frame i renders on stream i % depth into buffer i % depth, its exchange runs on the exchange stream
// behind an event, and the render of frame i + 1 (another stream) overlaps both the tail of render i and exchange i -- what
// toyraygun_amd/dist.py's DistributedRenderer does for one process per GPU, now behind the C ABI, so that a plain
// `bench.py --gpus N` is like for like with the pipelined N = 1 headline.  Re-use of a slot waits (on the device, by events) for the
// exchange that read it.  depth = 1 (default) is the progressive mode of the plugin: one buffer, the accumulation continued in place,
// the exchange on the render's stream.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
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
        GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(lib, "ncclGetVersion"));   // (the two below only feed trg_group_rccl_info: optional)
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
        return CommInitAll && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;
thread_local std::string g_group_create_err;

// every entry point may select devices on the CALLER's thread (its own HIP calls, or a group of one whose jobs run there): the device
// the caller had current is put back on the way out, so that e.g. a torch caller's next allocation does not land on the group's last device
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};
}  // namespace

struct trg_group {
    static constexpr int kMaxDepth = 8;
    int n = 0;
    uint32_t w = 0, h = 0, band = 0;          // band = rows per device = ceil(h / n)
    // TRG_BANDS_INTERLEAVED: device r renders the 8-row micro-bands r, r + n, ... (trg_render_bands) compactly into rows [r * stride, ...) of
    // its frame buffer; the exchange completes the COMPACT frame and trg_unpack_bands turns it into image[r][s] on the exchange stream
    int bands = TRG_BANDS_CONTIGUOUS;
    uint32_t stride = 0;                       // rows of a rank's slice of the compact frame: 8 * ceil(ceil(h / 8) / n)
    std::vector<std::vector<float *>> image;   // [r][s] w x h float4: the unpacked frame (interleaved bands only)
    std::vector<int> devices;
    std::vector<trg_ctx *> ctx;
    // frame[r][s]: device r's frame buffer of pipeline slot s -- n * band rows x w float4 (padded full frame); the slot a frame renders
    // into is bound as the context's accumulation buffer for that launch.  depth = 1: one buffer, bound once.
    int depth = 1;
    std::vector<std::vector<float *>> frame;
    std::vector<std::vector<hipStream_t>> rstream;   // [r][s] render streams of a pipelined group (depth > 1); depth 1: the context's current stream
    std::vector<hipStream_t> xstream;                // [r] exchange stream of a pipelined group; depth 1: the exchange rides on the render's stream
    std::vector<ncclComm_t> comm;
    bool copy_exchange = false;                // bands move by hipMemcpyPeerAsync + events instead of RCCL
    std::string fallback_note;                 // why the copy exchange was taken although RCCL was asked for (empty: it was not)
    // [r][s]: rank r's band of slot s is complete / rank r's part of the exchange of slot s (its copies of the others' bands, or its
    // side of the collective) is complete
    std::vector<std::vector<hipEvent_t>> rendered, taken;
    int slot_gather[kMaxDepth] = {}, slot_root[kMaxDepth] = {};   // the exchange the LAST frame rendered into slot s was followed by
    // RCCL exchange of a pipelined group: taken[r][s] of the LAST collective is recorded lazily, by rank r's own worker at the start of the next
    // trg_group_render (the collective is then still the last thing on xstream[r]) -- or by whoever needs it sooner (settle_taken)
    int taken_pending = -1;
    unsigned long long step = 0;               // trg_group_render calls so far
    int cur_slot = 0;                          // the slot of the last trg_group_render (what read_accum / the fences refer to)
    bool last_on_xstream = false;              // the last frame's final device work sits on the exchange streams (pipelined + gathered)
    hipEvent_t fence[8][64] = {};              // group fences [slot][rank] of a pipelined group (depth 1 uses the contexts' own fences)
    bool fence_set[8] = {};
    // per-launch timing without a host sync (trg_group_time_launches): event pairs around every trg_render, on the stream it runs on
    bool time_launches = false;
    std::vector<std::vector<hipEvent_t>> tev;  // [r]: e0, e1, e0, e1, ... (grow-only pool)
    std::vector<size_t> tused;                 // [r]: events of tev[r] recorded since the last read
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
    if (r >= 0) {
        const char *msg = trg_last_error(g->ctx[r]);
        return gfail(g, g->workers->rc(r), "device %d (rank %d): %s", g->devices[r], r, (msg && *msg) ? msg : "HIP call of the exchange failed");
    }
    return TRG_OK;
}

static hipStream_t render_stream(trg_group *g, int r, int slot) {
    return g->depth > 1 ? g->rstream[r][slot] : static_cast<hipStream_t>(trg::ctx_current_stream(g->ctx[r]));
}
static hipStream_t exchange_stream(trg_group *g, int r, int slot) { return g->depth > 1 ? g->xstream[r] : render_stream(g, r, slot); }

// RCCL exchange, pipelined: record the `taken` event of the last collective on rank r's exchange stream if that is still owed (taken_pending)
static hipError_t settle_taken_rank(trg_group *g, int r) {
    if (g->taken_pending < 0) return hipSuccess;
    return hipEventRecord(g->taken[r][g->taken_pending], g->xstream[r]);
}

// wait (host) for everything the group has enqueued on device r
static hipError_t sync_rank(trg_group *g, int r) {
    hipError_t e = hipSetDevice(g->devices[r]);
    if (e == hipSuccess && g->depth > 1) {
        for (int s = 0; s < g->depth && e == hipSuccess; ++s) e = hipStreamSynchronize(g->rstream[r][s]);
        if (e == hipSuccess) e = hipStreamSynchronize(g->xstream[r]);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(trg::ctx_current_stream(g->ctx[r])));
    return e;
}

// free the pipeline's extra buffers, streams and events of rank r (slot 0's frame stays: it is the group's base buffer)
static void drop_pipeline_rank(trg_group *g, int r) {
    (void)hipSetDevice(g->devices[r]);
    if (r < (int)g->frame.size())
        for (size_t s = 1; s < g->frame[r].size(); ++s)
            if (g->frame[r][s]) (void)hipFree(g->frame[r][s]);
    if (r < (int)g->frame.size() && !g->frame[r].empty()) g->frame[r].resize(1);
    if (r < (int)g->image.size()) {
        for (float *im : g->image[r]) if (im) (void)hipFree(im);
        g->image[r].clear();
    }
    if (r < (int)g->rstream.size()) {
        for (hipStream_t s : g->rstream[r]) if (s) (void)hipStreamDestroy(s);
        g->rstream[r].clear();
    }
    if (r < (int)g->xstream.size() && g->xstream[r]) { (void)hipStreamDestroy(g->xstream[r]); g->xstream[r] = nullptr; }
    for (auto *evs : { &g->rendered, &g->taken })
        if (r < (int)evs->size()) {
            for (hipEvent_t e : (*evs)[r]) if (e) (void)hipEventDestroy(e);
            (*evs)[r].clear();
        }
    for (int k = 0; k < 8; ++k)
        if (g->fence[k][r]) { (void)hipEventDestroy(g->fence[k][r]); g->fence[k][r] = nullptr; }
}

// the per-slot events of rank r for `depth` slots (created on r's device)
static hipError_t make_slot_events(trg_group *g, int r, int depth) {
    hipError_t e = hipSetDevice(g->devices[r]);
    g->rendered[r].assign((size_t)depth, nullptr); g->taken[r].assign((size_t)depth, nullptr);
    for (int s = 0; s < depth && e == hipSuccess; ++s) {
        e = hipEventCreateWithFlags(&g->rendered[r][s], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->taken[r][s], hipEventDisableTiming);
    }
    return e;
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
const char *trg_group_exchange_note(trg_group *g) { return g ? g->fallback_note.c_str() : ""; }
trg_ctx *trg_group_ctx(trg_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }
int trg_group_pipeline_depth(trg_group *g) { return g ? g->depth : 0; }

void trg_group_destroy(trg_group *g) {
    if (!g) return;
    DeviceGuard guard;
    g->workers.reset();   // joins the host threads (none of them has work: every entry point waits for its jobs)
    for (int r = 0; r < g->n; ++r)
        if (r < (int)g->ctx.size() && g->ctx[r]) (void)sync_rank(g, r);
    for (size_t r = 0; r < g->comm.size(); ++r)
        if (g->comm[r]) (void)g_rccl.CommDestroy(g->comm[r]);
    for (int r = 0; r < g->n; ++r) {
        if (r < (int)g->tev.size()) {
            (void)hipSetDevice(g->devices[r]);
            for (hipEvent_t e : g->tev[r]) if (e) (void)hipEventDestroy(e);
        }
        if (r < (int)g->ctx.size() && g->ctx[r]) {
            (void)trg_set_stream(g->ctx[r], nullptr);
            (void)trg_bind_accum(g->ctx[r], nullptr);
        }
        drop_pipeline_rank(g, r);
        if (r < (int)g->ctx.size() && g->ctx[r]) trg_destroy(g->ctx[r]);
        if (r < (int)g->frame.size() && !g->frame[r].empty() && g->frame[r][0]) { (void)hipSetDevice(g->devices[r]); (void)hipFree(g->frame[r][0]); }
    }
    (void)hipGetLastError();
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
    DeviceGuard guard;
    trg_group *g = new (std::nothrow) trg_group;
    if (!g) return gfail(nullptr, TRG_ERR_NOMEM, "trg_group_create: out of host memory");
    g->n = n; g->w = width; g->h = height; g->band = (height + (uint32_t)n - 1) / (uint32_t)n;
    g->devices.assign(devices, devices + n);
    g->ctx.assign(n, nullptr); g->frame.assign(n, std::vector<float *>(1, nullptr));
    g->rstream.assign(n, {}); g->xstream.assign(n, nullptr); g->rendered.assign(n, {}); g->taken.assign(n, {});
    g->tev.assign(n, {}); g->tused.assign(n, 0);
    trg_microband_rows(height, (uint32_t)n, 0, nullptr, &g->stride);
    g->image.assign(n, {});
    const size_t frame_bytes = (size_t)std::max(g->band, g->stride) * n * width * 16u;   // either band rule fits
    for (int r = 0; r < n; ++r) {
        int rc = trg_create(&g->ctx[r], devices[r], width, height);
        if (rc != TRG_OK) { gfail(nullptr, rc, "trg_group_create: device %d: %s", devices[r], trg_last_error(nullptr)); trg_group_destroy(g); return rc; }
        hipError_t e = hipSetDevice(devices[r]);
        if (e == hipSuccess) e = hipMalloc((void **)&g->frame[r][0], frame_bytes);
        if (e == hipSuccess) e = hipMemset(g->frame[r][0], 0, frame_bytes);
        if (e != hipSuccess) {
            gfail(nullptr, TRG_ERR_NOMEM, "trg_group_create: device %d: %s", devices[r], hipGetErrorString(e));
            trg_group_destroy(g);
            return TRG_ERR_NOMEM;
        }
        (void)trg_bind_accum(g->ctx[r], g->frame[r][0]);
    }
    // (TRG_GROUP_FORCE_RCCL: a group of ONE device goes through RCCL too -- dlopen, ncclCommInitAll, the in-place all-gather on the render's
    // stream -- so that as much of the exchange as one GPU can exercise is exercised by the GPU tests of a single-GPU box)
    if (!copy_exchange && (n > 1 || getenv("TRG_GROUP_FORCE_RCCL"))) {
        std::string err;
        if (!g_rccl.load(err)) {
            if (getenv("TRG_GROUP_FORCE_RCCL") || xenv) { gfail(nullptr, TRG_ERR_DEVICE, "trg_group_create: %s", err.c_str()); trg_group_destroy(g); return TRG_ERR_DEVICE; }
            copy_exchange = true;   // no librccl.so on this machine: the bands move by peer copies -- said so, not silently
            g->fallback_note = "RCCL unavailable (" + err + "): the bands are exchanged by hipMemcpyPeerAsync";
            fprintf(stderr, "toyraygun: %s\n", g->fallback_note.c_str());
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
    for (int r = 0; r < n; ++r) {
        hipError_t e = make_slot_events(g, r, 1);
        if (e != hipSuccess) { gfail(nullptr, TRG_ERR_DEVICE, "trg_group_create: device %d: %s", devices[r], hipGetErrorString(e)); trg_group_destroy(g); return TRG_ERR_DEVICE; }
        if (!g->copy_exchange) continue;
        for (int p = 0; p < n; ++p) {   // direct copies over xGMI where the devices can reach each other (otherwise the runtime stages them)
            int can = 0;
            if (devices[p] != devices[r] && hipDeviceCanAccessPeer(&can, devices[r], devices[p]) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(devices[p], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    (void)hipGetLastError();
    // the devices' host threads: started once, each bound to its device; a frame is handed to them through a condition variable.
    // (A group of one has no thread -- its jobs run on the caller's, where trg_render selects the device itself.)
    std::function<void(int)> on_start;
    if (n > 1) on_start = [g](int r) { (void)hipSetDevice(g->devices[r]); };
    try {
        g->workers.reset(new (std::nothrow) trg::DeviceWorkers(n, on_start));
    } catch (...) {   // std::thread could not start (std::system_error): no half-started group
        g->workers.reset();
    }
    if (!g->workers) { gfail(nullptr, TRG_ERR_NOMEM, "trg_group_create: out of host memory"); trg_group_destroy(g); return TRG_ERR_NOMEM; }
    // environment variable TRG_GROUP_BANDS=interleaved: the default band rule of every group of this process (the plugin's too)
    const char *benv = getenv("TRG_GROUP_BANDS");
    if (benv && strcmp(benv, "interleaved") == 0 && n > 1) {
        const int rc = trg_group_set_bands(g, TRG_BANDS_INTERLEAVED);
        if (rc != TRG_OK) { gfail(nullptr, rc, "trg_group_create: %s", g->err.c_str()); trg_group_destroy(g); return rc; }
    }
    *out = g;
    return TRG_OK;
}

int trg_group_sync(trg_group *g) {
    if (!g) return TRG_ERR_INVALID;
    DeviceGuard guard;
    return for_each_device(g, [&](int r) {
        const hipError_t e = sync_rank(g, r);
        return e == hipSuccess ? TRG_OK : TRG_ERR_DEVICE;
    });
}

int trg_group_set_pipeline(trg_group *g, int depth) {
    if (!g) return TRG_ERR_INVALID;
    if (depth < 1 || depth > trg_group::kMaxDepth) return gfail(g, TRG_ERR_INVALID, "trg_group_set_pipeline: depth must be 1..%d", trg_group::kMaxDepth);
    DeviceGuard guard;
    if (int rc = trg_group_sync(g)) return rc;
    const size_t frame_bytes = (size_t)std::max(g->band, g->stride) * g->n * g->w * 16u, image_bytes = (size_t)g->w * g->h * 16u;
    const int rc = for_each_device(g, [&](int r) {
        (void)trg_set_stream(g->ctx[r], nullptr);
        (void)trg_bind_accum(g->ctx[r], g->frame[r][0]);
        trg::ctx_forget_streams(g->ctx[r]);   // the group is drained (above) and its render streams are about to be destroyed
        drop_pipeline_rank(g, r);
        hipError_t e = make_slot_events(g, r, depth);
        if (g->bands == TRG_BANDS_INTERLEAVED && e == hipSuccess) {
            g->image[r].assign((size_t)depth, nullptr);
            for (int s = 0; s < depth && e == hipSuccess; ++s) {
                e = hipMalloc((void **)&g->image[r][s], image_bytes);
                if (e == hipSuccess) e = hipMemset(g->image[r][s], 0, image_bytes);
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (depth > 1 && e == hipSuccess) {
            g->frame[r].resize((size_t)depth, nullptr);
            g->rstream[r].assign((size_t)depth, nullptr);
            for (int s = 0; s < depth && e == hipSuccess; ++s) {
                if (s > 0) {
                    e = hipMalloc((void **)&g->frame[r][s], frame_bytes);
                    if (e == hipSuccess) e = hipMemset(g->frame[r][s], 0, frame_bytes);
                }
                if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->rstream[r][s], hipStreamNonBlocking);
            }
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->xstream[r], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipDeviceSynchronize();   // the memsets (null stream) before anything on a non-blocking stream touches the buffers
        }
        if (e != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_NOMEM; }
        // a pipelined group only enqueues (TRG_OPT_TIMING would wait for every launch), and it tells the context how many of its launches
        // overlap: separate per-stream scratch, frame split chosen for throughput
        if (depth > 1) (void)trg_set_option(g->ctx[r], TRG_OPT_TIMING, 0);
        return trg_set_option(g->ctx[r], TRG_OPT_LAUNCHES_IN_FLIGHT, depth);
    });
    if (rc != TRG_OK) {   // back to the one-buffer group rather than a half-built pipeline
        (void)for_each_device(g, [&](int r) { drop_pipeline_rank(g, r); (void)make_slot_events(g, r, 1); return TRG_OK; });
        g->depth = 1;
        return gfail(g, rc, "trg_group_set_pipeline: allocating %d frame slots per device failed", depth);
    }
    g->depth = depth;
    g->step = 0; g->cur_slot = 0; g->last_on_xstream = false; g->taken_pending = -1;
    memset(g->slot_gather, 0, sizeof(g->slot_gather));
    memset(g->fence_set, 0, sizeof(g->fence_set));
    return TRG_OK;
}

int trg_group_set_bands(trg_group *g, int mode) {
    if (!g) return TRG_ERR_INVALID;
    if (mode != TRG_BANDS_CONTIGUOUS && mode != TRG_BANDS_INTERLEAVED) return gfail(g, TRG_ERR_INVALID, "trg_group_set_bands: unknown mode %d", mode);
    if (mode == g->bands) return TRG_OK;
    g->bands = mode;
    const int rc = trg_group_set_pipeline(g, g->depth);   // (re)allocates the slots, with or without the unpacked images
    if (rc != TRG_OK) g->bands = TRG_BANDS_CONTIGUOUS;
    return rc;
}
int trg_group_bands(trg_group *g) { return g ? g->bands : 0; }

int trg_group_time_launches(trg_group *g, int on) {
    if (!g) return TRG_ERR_INVALID;
    g->time_launches = on != 0;
    return TRG_OK;
}

int trg_group_launch_ms(trg_group *g, int rank, double *ms_out, uint32_t cap, uint32_t *n_out) {
    if (!g || rank < 0 || rank >= g->n || !n_out) return TRG_ERR_INVALID;
    DeviceGuard guard;
    hipError_t e = sync_rank(g, rank);
    const size_t pairs = g->tused[rank] / 2;
    *n_out = (uint32_t)pairs;
    for (size_t k = 0; k < pairs && e == hipSuccess; ++k) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g->tev[rank][2 * k], g->tev[rank][2 * k + 1]);
        if (ms_out && k < cap) ms_out[k] = ms;
    }
    g->tused[rank] = 0;
    if (e != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_launch_ms: %s", hipGetErrorString(e));
    return TRG_OK;
}

int trg_group_load_scene(trg_group *g, const float *positions3, const float *normals3, const float *colors3, const uint32_t *indices,
                         const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris) {
    if (!g) return TRG_ERR_INVALID;
    DeviceGuard guard;
    if (int rc = trg_group_sync(g)) return rc;
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
    DeviceGuard guard;
    if (int rc = trg_group_sync(g)) return rc;
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
    DeviceGuard guard;
    if (int rc = trg_group_sync(g)) return rc;
    return for_each_device(g, [&](int r) { return trg_set_pixel_offsets_seed(g->ctx[r], seed); });
}
int trg_group_set_option(trg_group *g, int option, int64_t value) {
    if (!g) return TRG_ERR_INVALID;
    if (option == TRG_OPT_TIMING && value != 0 && g->depth > 1)
        return gfail(g, TRG_ERR_INVALID, "trg_group_set_option: TRG_OPT_TIMING waits for every launch -- a pipelined group (depth %d) measures with trg_group_time_launches", g->depth);
    for (int r = 0; r < g->n; ++r)
        if (int rc = trg_set_option(g->ctx[r], option, value)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
    return TRG_OK;
}

// is rank d a destination of exchange `gather` (to `root`)?
static bool is_destination(int gather, int root, int d) { return gather == TRG_GATHER_ALL || (gather == TRG_GATHER_ROOT && d == root); }

int trg_group_render(trg_group *g, uint32_t frameIndexBegin, uint32_t spp, uint32_t bounces, int gather, int root) {
    if (!g) return TRG_ERR_INVALID;
    if (gather != TRG_GATHER_NONE && gather != TRG_GATHER_ALL && gather != TRG_GATHER_ROOT) return gfail(g, TRG_ERR_INVALID, "trg_group_render: unknown gather mode %d", gather);
    if (gather == TRG_GATHER_ROOT && (root < 0 || root >= g->n)) return gfail(g, TRG_ERR_INVALID, "trg_group_render: root %d out of range", root);
    DeviceGuard guard;
    const int slot = (int)(g->step % (unsigned long long)g->depth);
    const int prev_gather = g->slot_gather[slot], prev_root = g->slot_root[slot];
    const bool exchange = gather != TRG_GATHER_NONE && (g->copy_exchange || !g->comm.empty());
    const bool il = g->bands == TRG_BANDS_INTERLEAVED;
    const size_t count = (size_t)(il ? g->stride : g->band) * g->w * 4u, bytes = count * sizeof(float);   // floats / bytes per rank's slice of the frame
    // ---- phase A, every device on its own host thread: wait (on the device) until the frame slot is free, render the band ----
    if (int rc = for_each_device(g, [&](int r) {
            hipStream_t rs = render_stream(g, r, slot);
            hipError_t e = hipSetDevice(g->devices[r]);   // (a worker is bound to its device already; a group of one runs here on the caller's thread)
            if (e == hipSuccess) e = settle_taken_rank(g, r);   // (the previous frame's collective: still the last thing on this rank's exchange stream)
            // the exchange that last read this slot must be over before the band is overwritten: the copies the destinations took of it
            // (copy exchange), this rank's side of the collective (RCCL).  Events of an earlier call: all recorded already.
            if (prev_gather != TRG_GATHER_NONE && e == hipSuccess) {
                if (g->copy_exchange) {
                    // (d == r: a destination's exchange stream also reads -- interleaved bands: unpacks -- its OWN band of frame[r][slot], and
                    //  records taken[r][slot] behind that; at depth 1 that work sits on the render's own stream and is ordered anyway)
                    for (int d = 0; d < g->n && e == hipSuccess; ++d)
                        if ((d != r || g->depth > 1) && is_destination(prev_gather, prev_root, d)) e = hipStreamWaitEvent(rs, g->taken[d][slot], 0);
                } else if (!g->comm.empty() && g->depth > 1) {
                    e = hipStreamWaitEvent(rs, g->taken[r][slot], 0);
                }
            }
            if (e != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_DEVICE; }
            if (g->depth > 1) {
                (void)trg_bind_accum(g->ctx[r], g->frame[r][slot]);
                (void)trg_set_stream(g->ctx[r], rs);
            }
            uint32_t row0, rows;
            trg_band_rows(g->h, (uint32_t)g->n, (uint32_t)r, &row0, &rows);
            hipEvent_t t0 = nullptr, t1 = nullptr;
            if (g->time_launches) {
                if (g->tused[r] + 2 > g->tev[r].size()) {
                    hipEvent_t a = nullptr, b = nullptr;
                    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_DEVICE; }
                    g->tev[r].push_back(a); g->tev[r].push_back(b);
                }
                t0 = g->tev[r][g->tused[r]]; t1 = g->tev[r][g->tused[r] + 1];
                if (hipEventRecord(t0, rs) != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_DEVICE; }
            }
            if (int rc = il ? trg_render_bands(g->ctx[r], frameIndexBegin, spp, bounces, (uint32_t)g->n, (uint32_t)r, (uint32_t)r * g->stride)
                            : trg_render(g->ctx[r], frameIndexBegin, spp, bounces, row0, rows))
                return rc;
            if (t1) {
                if (hipEventRecord(t1, rs) != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_DEVICE; }
                g->tused[r] += 2;
            }
            if (exchange && (g->copy_exchange || g->depth > 1)) e = hipEventRecord(g->rendered[r][slot], rs);
            // RCCL, pipelined: this rank's side of the collective goes to its exchange stream, behind its band -- ordered here, on the rank's own thread
            if (exchange && !g->copy_exchange && g->depth > 1 && e == hipSuccess) e = hipStreamWaitEvent(g->xstream[r], g->rendered[r][slot], 0);
            if (e != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_DEVICE; }
            // interleaved bands, and no exchange delivers a frame to this rank (none at all, or a gather to another root): its own micro-bands go
            // to their image rows right behind the render, so that trg_group_read_accum / _postprocess of this rank show its band either way
            if (il && (!exchange || !is_destination(gather, root, r))) return trg_unpack_bands(g->ctx[r], g->frame[r][slot], g->image[r][slot], (uint32_t)g->n);
            return TRG_OK;
        }))
        return rc;
    g->taken_pending = -1;   // (settled by every rank in phase A)
    g->slot_gather[slot] = exchange ? gather : TRG_GATHER_NONE; g->slot_root[slot] = root;
    g->cur_slot = slot; g->step++;
    g->last_on_xstream = exchange && g->depth > 1;
    if (!exchange) return TRG_OK;
    if (g->copy_exchange) {
        // ---- phase B (copy exchange), every DESTINATION on its own host thread: take band p from its owner once p has rendered it (the
        //      events of phase A: all recorded, the hand-off above is a barrier), on the destination's exchange stream.  The same
        //      hipMemcpyPeerAsync whether the two contexts sit on one device or on two.
        return for_each_device(g, [&](int d) {
            if (!is_destination(gather, root, d)) return TRG_OK;
            hipStream_t xs = exchange_stream(g, d, slot);
            hipError_t e = hipSuccess;
            if (g->depth > 1) e = hipStreamWaitEvent(xs, g->rendered[d][slot], 0);   // (its own band: keeps `taken` behind the whole frame)
            for (int p = 0; p < g->n && e == hipSuccess; ++p) {
                if (p == d) continue;
                e = hipStreamWaitEvent(xs, g->rendered[p][slot], 0);
                if (e == hipSuccess)
                    e = hipMemcpyPeerAsync(g->frame[d][slot] + (size_t)p * count, g->devices[d], g->frame[p][slot] + (size_t)p * count, g->devices[p], bytes, xs);
            }
            if (e != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_DEVICE; }
            if (il) {   // the compact frame is complete on this destination: to image rows, on the exchange stream, before `taken`
                (void)trg_set_stream(g->ctx[d], xs);
                const int urc = trg_unpack_bands(g->ctx[d], g->frame[d][slot], g->image[d][slot], (uint32_t)g->n);
                (void)trg_set_stream(g->ctx[d], g->depth > 1 ? render_stream(g, d, slot) : xs);
                if (urc != TRG_OK) return urc;
            }
            e = hipEventRecord(g->taken[d][slot], xs);
            if (e != hipSuccess) { (void)hipGetLastError(); return TRG_ERR_DEVICE; }
            return TRG_OK;
        });
    }
    // ---- phase B (RCCL): the one exchange of the frame, enqueued behind each device's render -- on the render's own stream (depth 1:
    //      the context's current stream, trg_set_stream's if the caller set one through trg_group_ctx), or on the device's exchange
    //      stream, which phase A has put behind the `rendered` event (pipelined).  One grouped call from this thread for all devices (the
    //      single-process, several-devices pattern of the NCCL documentation) and NOTHING else: the event waits in front of it were issued by
    //      the ranks' own threads in phase A, the `taken` records behind it are issued by them at the start of the next frame (taken_pending).
    ncclResult_t nr = g_rccl.GroupStart();
    for (int r = 0; r < g->n && nr == ncclSuccess; ++r) {
        hipStream_t s = exchange_stream(g, r, slot);
        float *mine = g->frame[r][slot] + (size_t)r * count;
        if (gather == TRG_GATHER_ALL) {
            nr = g_rccl.AllGather(mine, g->frame[r][slot], count, ncclFloat, g->comm[r], s);   // in place: sendbuff = recvbuff + rank * count
        } else if (r == root) {
            for (int p = 0; p < g->n && nr == ncclSuccess; ++p)
                if (p != root) nr = g_rccl.Recv(g->frame[r][slot] + (size_t)p * count, count, ncclFloat, p, g->comm[r], s);
        } else {
            nr = g_rccl.Send(mine, count, ncclFloat, root, g->comm[r], s);
        }
    }
    const ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_render: RCCL: %s", g_rccl.GetErrorString(nr));
    if (il) {
        // the compact frame is complete where the exchange delivered it: to image rows, behind the collective on the same stream -- one launch
        // from this thread for a gather to one root, one per rank on the ranks' threads for a gather to all
        auto unpack = [&](int r) {
            hipStream_t xs = exchange_stream(g, r, slot);
            (void)trg_set_stream(g->ctx[r], xs);
            const int urc = trg_unpack_bands(g->ctx[r], g->frame[r][slot], g->image[r][slot], (uint32_t)g->n);
            (void)trg_set_stream(g->ctx[r], g->depth > 1 ? render_stream(g, r, slot) : xs);
            return urc;
        };
        if (gather == TRG_GATHER_ROOT || g->n == 1) {
            const int r = gather == TRG_GATHER_ROOT ? root : 0;
            if (int rc = unpack(r)) return gfail(g, rc, "device %d (rank %d): %s", g->devices[r], r, trg_last_error(g->ctx[r]));
        } else if (int rc = for_each_device(g, [&](int r) { return unpack(r); })) {
            return rc;
        }
    }
    if (g->depth > 1) g->taken_pending = slot;
    return TRG_OK;
}

// frames in flight (MetalRenderer.mm:33,377,385-387: a semaphore of three): a mark behind the LAST frame on every device -- on the stream
// its last device work went to (the render's, or the exchange stream of a pipelined group) -- and the wait for all of them
int trg_group_fence_record(trg_group *g, int slot) {
    if (!g || slot < 0 || slot >= 8) return TRG_ERR_INVALID;
    DeviceGuard guard;
    if (g->depth == 1) {
        for (int r = 0; r < g->n; ++r)
            if (int rc = trg_fence_record(g->ctx[r], slot)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
        return TRG_OK;
    }
    for (int r = 0; r < g->n; ++r) {
        hipError_t e = hipSetDevice(g->devices[r]);
        if (e == hipSuccess && !g->fence[slot][r]) e = hipEventCreateWithFlags(&g->fence[slot][r], hipEventDisableTiming);
        // the last frame's device work: its render stream, and -- behind an exchange -- the exchange stream too (a rank that is no destination
        // of a gather to one root unpacks its own micro-bands on the render stream AFTER the `rendered` event the exchange stream waits for):
        // the mark goes behind both
        if (e == hipSuccess) e = hipEventRecord(g->fence[slot][r], g->rstream[r][g->cur_slot]);
        if (e == hipSuccess && g->last_on_xstream) {
            e = hipStreamWaitEvent(g->xstream[r], g->fence[slot][r], 0);
            if (e == hipSuccess) e = hipEventRecord(g->fence[slot][r], g->xstream[r]);
        }
        if (e != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_fence_record: rank %d: %s", r, hipGetErrorString(e));
    }
    g->fence_set[slot] = true;
    return TRG_OK;
}
int trg_group_fence_wait(trg_group *g, int slot) {
    if (!g || slot < 0 || slot >= 8) return TRG_ERR_INVALID;
    DeviceGuard guard;
    if (g->depth == 1) {
        for (int r = 0; r < g->n; ++r)
            if (int rc = trg_fence_wait(g->ctx[r], slot)) return gfail(g, rc, "rank %d: %s", r, trg_last_error(g->ctx[r]));
        return TRG_OK;
    }
    if (!g->fence_set[slot]) return TRG_OK;
    for (int r = 0; r < g->n; ++r) {
        hipError_t e = hipSetDevice(g->devices[r]);
        if (e == hipSuccess && g->fence[slot][r]) e = hipEventSynchronize(g->fence[slot][r]);
        if (e != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_fence_wait: rank %d: %s", r, hipGetErrorString(e));
    }
    return TRG_OK;
}

int trg_group_read_accum(trg_group *g, int rank, float *rgba) {
    if (!g || rank < 0 || rank >= g->n || !rgba) return TRG_ERR_INVALID;
    DeviceGuard guard;
    if (sync_rank(g, rank) != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_read_accum: rank %d: synchronise failed", rank);
    if (g->bands == TRG_BANDS_INTERLEAVED) {   // the unpacked frame of the last trg_group_render
        hipError_t e = hipSetDevice(g->devices[rank]);
        if (e == hipSuccess) e = hipMemcpy(rgba, g->image[rank][g->cur_slot], (size_t)g->w * g->h * 16u, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_read_accum: rank %d: %s", rank, hipGetErrorString(e));
        return TRG_OK;
    }
    // the frame of the last trg_group_render: its slot is what the context is bound to
    if (int rc = trg_read_accum(g->ctx[rank], rgba)) return gfail(g, rc, "rank %d: %s", rank, trg_last_error(g->ctx[rank]));
    return TRG_OK;
}

// N1 for a group: tonemap the last frame of device `rank` (the unpacked image when the bands are interleaved)
int trg_group_postprocess(trg_group *g, int rank, uint8_t *rgba8, int flip_y) {
    if (!g || rank < 0 || rank >= g->n || !rgba8) return TRG_ERR_INVALID;
    DeviceGuard guard;
    if (sync_rank(g, rank) != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_postprocess: rank %d: synchronise failed", rank);
    trg_ctx *c = g->ctx[rank];
    const bool il = g->bands == TRG_BANDS_INTERLEAVED;
    if (il) (void)trg_bind_accum(c, g->image[rank][g->cur_slot]);
    const int rc = trg_postprocess(c, rgba8, flip_y);
    if (il) (void)trg_bind_accum(c, g->frame[rank][g->cur_slot]);
    if (rc != TRG_OK) return gfail(g, rc, "rank %d: %s", rank, trg_last_error(c));
    return TRG_OK;
}

int trg_group_rccl_info(trg_group *g, int *version, int *ranks_out, int cap) {
    if (!g) return TRG_ERR_INVALID;
    if (version) *version = 0;
    if (g->comm.empty()) return gfail(g, TRG_ERR_INVALID, "trg_group_rccl_info: this group has no RCCL communicator (%s)", g->copy_exchange ? "copy exchange" : "one device");
    if (!g_rccl.GetVersion || !g_rccl.CommCount) return gfail(g, TRG_ERR_DEVICE, "trg_group_rccl_info: librccl.so lacks ncclGetVersion / ncclCommCount");
    int v = 0;
    ncclResult_t nr = g_rccl.GetVersion(&v);
    if (version) *version = v;
    for (int r = 0; r < g->n && r < cap && nr == ncclSuccess && ranks_out; ++r) nr = g_rccl.CommCount(g->comm[r], &ranks_out[r]);
    if (nr != ncclSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_rccl_info: %s", g_rccl.GetErrorString(nr));
    return TRG_OK;
}

int trg_group_get_stats(trg_group *g, trg_stats *out) {
    if (!g || !out) return TRG_ERR_INVALID;
    DeviceGuard guard;
    memset(out, 0, sizeof(*out));
    for (int r = 0; r < g->n; ++r) {
        trg_stats st;
        // (the ray counters are added to by every launch on every render stream: all of them first)
        if (g->depth > 1 && sync_rank(g, r) != hipSuccess) return gfail(g, TRG_ERR_DEVICE, "trg_group_get_stats: rank %d: synchronise failed", r);
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
