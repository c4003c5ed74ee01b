// trg_capi.cpp -- the C ABI of include/trg.h: context, scene upload + BVH build, launches, stats.
// Host-only C++ (HIP runtime API); the kernels live in trg_kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/trg.h"
#include "bvh_build.h"
#include "q4node.h"
#include "q8node.h"
#include "trg_build.h"
#include "trg_internal.h"
#include "trg_kernels.h"

using namespace trg;
// the host-only debug entry points build WITHOUT box leaves unless TRG_DEBUG_BVH_BOXES is set (tests: the builder's soundness checks run both ways)
static bool debug_want_boxes() { const char *e = getenv("TRG_DEBUG_BVH_BOXES"); return e && atoi(e) != 0; }

struct trg_ctx {
    int device = 0;
    uint32_t w = 0, h = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t fence[8] = {};
    bool fence_set[8] = {};
    float *accum_own = nullptr, *accum = nullptr;
    uint32_t *offsets = nullptr;
    unsigned long long *counters = nullptr;
    unsigned char *blob = nullptr;
    unsigned char *tex_mem = nullptr;   // uv | ids | table | texels (trg_load_textures)
    TexDesc tex{};
    // global overflow levels of the traversal stacks (grow-only), one buffer per launch the caller keeps in flight
    // (TRG_OPT_LAUNCHES_IN_FLIGHT): launch k uses slot k mod in_flight, so overlapping launches never share one
    static constexpr int kScratchSlots = 16;
    hipStream_t slot_stream[kScratchSlots] = {};   // which stream owns scratch slot k (slot 0 = the context's own stream)
    int slots_used = 1;
    int *stack_scratch[kScratchSlots] = {};
    size_t stack_scratch_bytes[kScratchSlots] = {};
    // wavefront schedule: path state + ray queues of one batch, one set per launch in flight (grow-only)
    unsigned char *wf_mem[kScratchSlots] = {};
    size_t wf_bytes[kScratchSlots] = {};
    int cu_count = 256;
    uint32_t *xq = nullptr;   // kScratchSlots x 8 job-queue heads of the persistent regeneration launches (TRG_OPT_TILE_ORDER 64 + n), one set per stream
    uint32_t bvh_depth4 = 0, bvh_nodes4 = 0;
    SceneDesc sc{};
    bool scene_loaded = false, have_uniforms = false, have_offsets = false;
    trg_uniforms u{};
    bool opt_strict = false, opt_counters = false, opt_force_global = false, opt_timing = true;
    int opt_kernel = TRG_KERNEL_AUTO;
    uint32_t last_kernel = TRG_KERNEL_DIRECT;
    uint32_t last_tail_k = 0;
    int opt_gpu_build = 0;   // TRG_OPT_GPU_BUILD: 0 host SAH, 1 device binned SAH, 2 device LBVH (Karras), 3 device PLOC
    int opt_fsplit = 0;  // 0 = auto
    int opt_tail = -1;   // TRG_OPT_TAIL_BOUNCE: -1 auto, 0 off, K
    int opt_regen = -1;        // TRG_OPT_REGEN: 1 = path regeneration for HBM-resident scenes (direct kernel, frame-serial), -1 = from 32,768 triangles on, 0 = the lock-step kernel
    uint32_t last_regen = 0;
    int opt_tail_levels = 0;   // TRG_OPT_TAIL_LEVELS: 0 = re-compact every second bounce after K, 1 = once at K only
    int opt_in_flight = 1;  // launches of this context the caller keeps in flight (TRG_OPT_LAUNCHES_IN_FLIGHT)
    int opt_stack_levels = (int)TRG_STACK_LDS_LEVELS;   // TRG_OPT_STACK_LDS_LEVELS
    int opt_tail_sort = 0;     // TRG_OPT_TAIL_SORT: 0 off, 1 direction octant, 2 / 3 octant + origin cell of a 2^3 / 4^3 grid
    int opt_tail_refill = -1;  // TRG_OPT_TAIL_REFILL: 1 = the tail launches run ONE bounce each with in-wave refill (render_rtail_kernel); -1 = the library's choice
    float scene_lo[3] = { 0.f, 0.f, 0.f }, scene_hi[3] = { 1.f, 1.f, 1.f };   // bounds of the loaded scene (tail sort: the origin grid)
    int opt_tile_order = -1;   // TRG_OPT_TILE_ORDER: -1 auto, 0 image columns centre-out, 1 / 2 / 4 / 8 XCD regions with that many column strips
    uint32_t last_xcd_cols = 0;
    double last_build_ms = 0.0;
    bool gpu_built = false;
    uint32_t bvh_nodes = 0, bvh_depth = 0, bvh_leaves = 0, bvh_quads = 0, bvh_boxes = 0;
    double last_ms = 0.0, total_ms = 0.0;
    uint32_t renders = 0;
    uint32_t last_fsplit = 1;
    uint32_t launches = 0;   // trg_render launches since create (never reset: picks the scratch slot)
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(trg_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}
#define HIPCHK(c, expr)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail((c), TRG_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }
static inline uint64_t align16_64(uint64_t v) { return (v + 15ull) & ~15ull; }

// Byte layout of the scene blob (DESIGN.md "Data layout in HBM"), computed in 64 bits: SceneDesc keeps 32-bit offsets, so a
// scene whose blob would not fit below 4 GiB is refused BEFORE anything is allocated or copied (the sums used to be
// 32-bit and wrapped from about 30 M triangles on).  The LDS part (scenes small enough to be staged; empty otherwise):
// n_nodes BVH2 / sign-ordered nodes, nt_rec 48-byte triangle records, attr_tris triangles with attribute arrays, the Halton
// tables.  Then, for every scene, what the HBM kernels traverse: n_nodes4 quantised wide nodes and n_fat 128-byte leaf records.
constexpr uint64_t kBlobLimit = 0xFFFFFFF0ull;
static bool plan_scene_layout(uint64_t n_nodes, uint32_t node_bytes, uint64_t nt_rec, uint64_t attr_tris, bool with_htab, uint64_t n_nodes4, uint64_t n_fat,
                              SceneDesc &sc, uint64_t &total, uint64_t n_nodes4_box = 0, uint64_t n_boxrec = 0) {
    const uint64_t off_nodes = 0;
    const uint64_t off_tris = align16_64(off_nodes + n_nodes * node_bytes);
    const uint64_t off_normals = align16_64(off_tris + nt_rec * 48ull);
    const uint64_t off_colors = align16_64(off_normals + attr_tris * 36ull);
    const uint64_t off_mats = align16_64(off_colors + attr_tris * 36ull);
    const uint64_t off_meta = align16_64(off_mats + attr_tris * 4ull);                  // u16 per record (before the Halton tables: the pool kernel stages up to them)
    const uint64_t off_htab = align16_64(off_meta + nt_rec * 2ull);
    const uint64_t stage_end = align16_64(off_htab + (with_htab ? kHtabBytes : 0u));
    const uint64_t off_nodes4 = (stage_end + 127ull) & ~127ull;  // 64-byte nodes, two per 128-byte line
    // (the box-leaf flavour of the 4-wide nodes -- only a host build of a scene with boxes has it -- sits BEFORE the leaf records: the kernels
    //  address records by an unsigned 32-bit distance from the node array they traverse)
    const uint64_t off_nodes4_box = (off_nodes4 + n_nodes4 * (TRG_WIDE8 ? kQ8NodeBytes : kQ4NodeBytes) + 127ull) & ~127ull;
    const uint64_t off_fat = (off_nodes4_box + n_nodes4_box * kQ4NodeBytes + 127ull) & ~127ull;   // one record per 128-byte line
    const uint64_t off_fat_planes = off_fat + n_fat * kFatRecBytes;                        // the shipped build's leaf records (planes), one per line as well
    const uint64_t off_boxrec = off_fat_planes + n_fat * kFatRecBytes;                     // ... and, right behind them at the SAME stride, the box records: box b is "record n_fat + b"
    const uint64_t off_tris_alt = off_boxrec + n_boxrec * kFatRecBytes;                    // the plane records of an LDS-sized scene (16-byte aligned)
    const uint64_t off_flat = (off_tris_alt + nt_rec * 48ull + 63ull) & ~63ull;            // the flat primitive list of a tiny scene (sc.n_flat set by the caller)
    total = off_flat + (uint64_t)sc.n_flat * kFlatPrimBytes + 128ull;
    sc.off_flat = (uint32_t)off_flat;
    if (total > kBlobLimit) return false;
    sc.off_nodes4_box = n_nodes4_box ? (uint32_t)off_nodes4_box : (uint32_t)off_nodes4; sc.off_boxrec = (uint32_t)off_boxrec; sc.n_boxrec = n_nodes4_box ? (uint32_t)n_boxrec : 0u;
    sc.off_meta = (uint32_t)off_meta; sc.off_tris_alt = (uint32_t)off_tris_alt;
    sc.off_nodes = (uint32_t)off_nodes; sc.off_tris = (uint32_t)off_tris; sc.off_normals = (uint32_t)off_normals;
    sc.off_colors = (uint32_t)off_colors; sc.off_mats = (uint32_t)off_mats; sc.off_htab = (uint32_t)off_htab;
    sc.off_nodes4 = (uint32_t)off_nodes4;
    sc.off_fat = (uint32_t)off_fat; sc.n_fat = (uint32_t)n_fat; sc.off_fat_planes = (uint32_t)off_fat_planes;
    sc.lds_stage_bytes = with_htab ? (uint32_t)stage_end : 0u;
    sc.blob_bytes = (uint32_t)total;
    return true;
}

// One 128-byte leaf record (trg_device.h kRecV4): rows 0..2 = the 48-byte geometry record (v0 | original index, e1 | mask = material id,
// e2 | 0), then the triangle's nine normal and nine colour floats (Raytracing.metal:104-108 reads them as attributes[triangle*3 + corner]).
static void fill_fat_record(unsigned char *dst, const F4 *rec48, const float *nrm, const float *col, uint32_t n_tris) {
    float *o = reinterpret_cast<float *>(dst);
    memcpy(o, rec48, 48);
    uint32_t prim;
    memcpy(&prim, &rec48[0].w, 4);
    if (prim < n_tris) {
        memcpy(o + 12, nrm + (size_t)prim * 9, 36);
        memcpy(o + 21, col + (size_t)prim * 9, 36);
    } else {
        memset(o + 12, 0, 72);
    }
    o[30] = 0.0f; o[31] = 0.0f;
}

// The PLANE form of one triangle (trg_device.h tri_test_planes; shipped build, LDS-resident scenes): from the same fp32 (v0, e1, e2) the
// Moeller-Trumbore record holds, in double:  n = e1 x e2 / |e1 x e2|, d0 = n . v0  (the supporting plane: t = (d0 - n.o) / (n.dir));
// n1 = (e2 x n) / (e1 . (e2 x n)), d1 = -n1 . v0  (u = n1 . P + d1 = weight of vertex 1);  n2 = (n x e1) / (e2 . (n x e1)), d2 = -n2 . v0
// (v = weight of vertex 2).  A degenerate triangle gets planes no ray passes (u = -1).  meta = (original index << 2) | (material id & 3):
// rays carry mask 3 (primary) or 1 (secondary), so the two low bits of the material id are all the mask test ever sees.
static void fill_plane_record(unsigned char *dst, uint16_t *meta, const F4 *rec48, const float *center, const F4 *quad_y = nullptr) {
    // (relative to `center`: SceneDesc::center -- the ray origin is shifted by it too, trg_device.h trav_begin)
    // quad_y: this record is the X of a QUAD leaf (bvh_build.h) and quad_y its Y record: the planes are the parallelogram's --
    // P = p0 + s X.e1 + t Y.e2 -- so that ONE test decides both triangles (s >= t: X with weights (s - t, t); t > s: Y with (s, t - s))
    const F4 &second = quad_y ? quad_y[2] : rec48[2];
    const double v0[3] = { (double)rec48[0].x - center[0], (double)rec48[0].y - center[1], (double)rec48[0].z - center[2] }, e1[3] = { rec48[1].x, rec48[1].y, rec48[1].z }, e2[3] = { second.x, second.y, second.z };
    auto cross = [](const double *a, const double *b, double *o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    double n[3], a1[3], a2[3];
    cross(e1, e2, n);
    const double len = std::sqrt(dot(n, n));
    float o[12] = { 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, -1.f };   // never hit
    if (len > 0.0 && std::isfinite(len)) {
        for (int k = 0; k < 3; ++k) n[k] /= len;
        cross(e2, n, a1); cross(n, e1, a2);
        const double s1 = dot(e1, a1), s2 = dot(e2, a2);
        if (s1 != 0.0 && s2 != 0.0) {
            for (int k = 0; k < 3; ++k) { a1[k] /= s1; a2[k] /= s2; }
            const double d0 = dot(n, v0), d1 = -dot(a1, v0), d2 = -dot(a2, v0);
            const float r[12] = { (float)n[0], (float)n[1], (float)n[2], (float)d0, (float)a1[0], (float)a1[1], (float)a1[2], (float)d1,
                                  (float)a2[0], (float)a2[1], (float)a2[2], (float)d2 };
            bool ok = true;
            for (float f : r) ok = ok && std::isfinite(f);
            if (ok) memcpy(o, r, sizeof(o));
        }
    }
    memcpy(dst, o, 48);
    uint32_t prim, mask;
    memcpy(&prim, &rec48[0].w, 4); memcpy(&mask, &rec48[1].w, 4);
    *meta = (uint16_t)((prim << 2) | (mask & 3u));
}

// The shipped build's 128-byte leaf record: rows 0..2 the three planes (fill_plane_record), then the original index and the material id
// (floats 12, 13), then the nine normal and nine colour floats (14..31).
static void fill_fat_record_planes(unsigned char *dst, const F4 *rec48, const float *nrm, const float *col, uint32_t n_tris, const float *center, const F4 *quad_y) {
    fill_fat_record(dst, rec48, nrm, col, n_tris);
    uint16_t meta;
    fill_plane_record(dst, &meta, rec48, center, quad_y);
    if (TRG_REC_META_FIRST) {   // index and mask right behind the planes (the first 64 bytes are all a test reads), the attributes at floats 14..31
        memmove(dst + 56, dst + 48, 72);
        memcpy(dst + 48, &rec48[0].w, 4);   // float 12: original index
        memcpy(dst + 52, &rec48[1].w, 4);   // float 13: material id = mask
    } else {
        memcpy(dst + 120, &rec48[0].w, 4);   // float 30: original index
        memcpy(dst + 124, &rec48[1].w, 4);   // float 31: material id = mask
    }
}

constexpr uint32_t kStackLdsLevels = TRG_STACK_LDS_LEVELS;
#ifndef TRG_TAIL_REFILL_AUTO
#define TRG_TAIL_REFILL_AUTO 0   // what TRG_OPT_TAIL_REFILL -1 means (set once the refill tail has been measured)
#endif
#ifndef TRG_TAIL_AUTO_MIN_BOUNCES
#define TRG_TAIL_AUTO_MIN_BOUNCES 4
#endif
#ifndef TRG_TAIL_AUTO_K
#define TRG_TAIL_AUTO_K 2
#endif
#ifndef TRG_TAIL_LEVEL_STEP
#define TRG_TAIL_LEVEL_STEP 2
#endif
constexpr uint32_t kRegenAutoMinTris = 32768u;   // TRG_OPT_REGEN -1: path regeneration from this many triangles on
constexpr uint32_t kTailAutoMinBounces = TRG_TAIL_AUTO_MIN_BOUNCES, kTailAutoK = TRG_TAIL_AUTO_K, kTailChunkFrames = 16, kTailLevelStep = TRG_TAIL_LEVEL_STEP;
#ifndef TRG_XCD_AUTO_COLS
#define TRG_XCD_AUTO_COLS 0   // TRG_OPT_TILE_ORDER -1 for scenes in HBM: 0 = image columns (measured faster, see choose_xcd_cols), 2 = XCD-aware 2 x 4
#endif

struct LdsPlan { bool lds_scene; uint32_t stack_off, red_off, pool_off, acc_off, total, klds, overflow_levels; };
static bool plan_lds_as(const trg_ctx *c, LdsPlan &p, bool lds_scene, bool pool, uint32_t fp_slots, uint32_t &limit, bool park = true) {
    p.lds_scene = lds_scene;
    uint32_t levels;
    if (p.lds_scene) {
        // the sentinel at level 0 + BVH2, near child first: at most one pending entry per level + the scratch slot above the top
        // (BVH2: leaves sit at depth <= bvh_depth, so inner nodes at depths 0 .. bvh_depth - 1 hold at most bvh_depth pending entries in levels
        //  1 .. bvh_depth, and the step at an inner node of depth k stores its far child at level <= k + 1: bvh_depth + 1 levels are used, one is spare)
        levels = c->bvh_depth + 2;
        p.klds = levels;
    } else {
        // the sentinel at level 0; 4-wide: up to three pending entries per level.  (TRG_WIDE8: one two-word node group per level of the 8-wide tree,
        // behind two sentinel levels: LdsStackT::first)
        levels = TRG_WIDE8 ? 2 * c->bvh_depth4 + 4 : 3 * c->bvh_depth4 + 3;
        p.klds = std::min(levels, (uint32_t)c->opt_stack_levels);
    }
    p.overflow_levels = levels - p.klds;
    // the pool kernel does not use the Halton group tables at the end of the staged region: it stages (and pays for) less
    p.stack_off = p.lds_scene ? align16(pool ? c->sc.off_htab : c->sc.lds_stage_bytes) : 0u;
    p.red_off = p.stack_off + p.klds * kBlock * 4u;
    p.pool_off = align16(p.red_off + 4u * 8u * 4u);
    p.total = p.pool_off;
#if TRG_EXPERIMENTS
    if (pool) {
        const uint32_t slots = (uint32_t)kBlock * (uint32_t)kPoolS;
        p.total = p.pool_off + slots * kPoolSlotBytes + 4u * slots * 2u + 16u;  // slots, two lists of 2P u16, counters
    }
#endif
    if (fp_slots) p.total = p.pool_off + fp_slots * (uint32_t)kBlock * 12u;  // render_fp_kernel: parked radiances, 3 floats x 256 pixel-frames per slot group
    // render_kernel on an HBM-resident scene parks the running average in LDS between frames (three VGPRs less across every
    // traversal; an LDS-resident scene has neither the room -- 8 workgroups of 20 KB per CU -- nor the need: it is spill-free)
    p.acc_off = 0;
    if (park && !p.lds_scene && !pool && !fp_slots) { p.acc_off = p.total; p.total += (uint32_t)kBlock * (TRG_PARK_PATH ? (TRG_PARK_OFFSET ? 40u : 36u) : 12u); }
    // render_fp_kernel on an HBM-resident scene parks throughput, radiance and the Halton offset the same way (seven words per thread; its
    // running average stays in the registers of the one wave per sub-tile that folds)
    if (park && !p.lds_scene && !pool && fp_slots && TRG_PARK_PATH && TRG_PARK_OFFSET) { p.acc_off = p.total; p.total += (uint32_t)kBlock * 28u; }
    limit = (pool || fp_slots) ? 160u * 1024u : 64u * 1024u;  // above 64 KB the launcher opts in per kernel
    return p.total <= limit;
}
// the frame-serial plan without touching the context's error state: true when there is one
static bool plan_lds_quiet(const trg_ctx *c, LdsPlan &p) {
    const bool want_lds = !c->opt_force_global && c->sc.lds_stage_bytes != 0 && c->sc.lds_stage_bytes <= kMaxLdsScene;
    uint32_t limit = 0;
    if (want_lds && plan_lds_as(c, p, true, false, 0, limit, true)) return true;
    return plan_lds_as(c, p, false, false, 0, limit, true);
}
static int plan_lds(trg_ctx *c, LdsPlan &p, bool pool = false, uint32_t fp_slots = 0, bool park = true) {
    const bool want_lds = !c->opt_force_global && c->sc.lds_stage_bytes != 0 && c->sc.lds_stage_bytes <= kMaxLdsScene;
    uint32_t limit = 0;
    if (want_lds && plan_lds_as(c, p, true, pool, fp_slots, limit, park)) return TRG_OK;
    // the scene stays in HBM -- also a small one whose tree is too deep for its traversal stacks to fit in LDS next to it
    if (plan_lds_as(c, p, false, pool, fp_slots, limit, park)) return TRG_OK;
    return fail(c, TRG_ERR_RANGE, "BVH depth %u needs %u B of LDS per workgroup (limit %u)", c->bvh_depth, p.total, limit);
}

// How many wavefronts share a pixel's frames (render_fp_kernel) for a launch of `rows` rows and `spp` frames.
// render_kernel walks the frames of a pixel serially in one lane: best while the 16x16-tile grid oversubscribes
// the chip many times over (kResidentGroups workgroups fit at once), wasteful when it does not -- the launch then
// ends with CUs idle behind a few workgroups' serial chains.  Splitting the frames over 4 (2) waves multiplies the
// workgroups and divides the chain.  Measured on C2 row bands (profiles/r01/fsplit.md): 4 lanes win below ~4
// resident sets (1/2, 1/4, 1/8 of a 1080p frame: -8 %, -18 %, -36 %) and lose above (full 1080p +1.6 %, 4K +9 %:
// every workgroup stages the scene and folds once more per pixel).
constexpr uint32_t kFpMaxRounds = 4;         // parked rounds per fold: 3 KB of LDS per round
constexpr uint32_t kFpMaxRoundsHbm = 2;      // ... for a scene traversed from HBM (render_fp_kernel<false>: 6 waves/SIMD = 6 workgroups of 26 KB per CU)
constexpr uint32_t kResidentGroups = 1536;   // 256 CUs x 6 workgroups of 4 waves (the HBM kernels' 6 waves/SIMD; LDS scenes: 8)
static uint32_t choose_fsplit(const trg_ctx *c, uint32_t spp, uint32_t rows) {
    if (c->opt_fsplit == 1 || spp < 2) return 1u;
    if (c->opt_fsplit == 2 || c->opt_fsplit == 4) return (uint32_t)c->opt_fsplit;
    const uint64_t groups = (uint64_t)((c->w + kTileW - 1) / kTileW) * ((rows + kTileH - 1) / kTileH);
    // a large scene in HBM is rendered by the path-regeneration kernel, which takes the frame lanes itself (so many workgroups per
    // tile).  One launch alone (C4, 16 spp, ms; lock step with 4 lanes / regeneration with 1, 2, 4 lanes; scripts/gpu_c4_bands.py):
    // 135 rows 4.1 / 6.1 3.96 3.74, 270 rows 7.2 / 7.2 6.96 6.07, 540 rows 13.4 / 12.8 11.0 11.2, 1080 rows 24.1 / 20.0 19.0 19.7;
    // four full frames in flight: 17.8, 18.4, 19.6 per step -- the overlap hides the ends of the pools better than smaller pools do
    // (row bands with four launches in flight, ms per step for 1 / 2 / 4 lanes: 135 rows 3.06 2.86 2.94, 270 rows 5.42 5.43 5.63, 540 rows
    // 10.1 10.2 10.8: scripts/gpu_c4_bands_pipe.py)
    if (c->opt_regen != 0 && c->sc.n_tris >= kRegenAutoMinTris) {
        if (c->opt_in_flight < 2) return groups > 3000ull ? 2u : (spp >= 4 ? 4u : 2u);
        return groups * (uint64_t)c->opt_in_flight < 4ull * kResidentGroups ? 2u : 1u;
    }
    // a caller that overlaps consecutive launches (k frames in flight on k streams) hides the tail by itself: then the
    // frame-serial kernel wins as soon as the launches in flight together hold about two resident sets of workgroups
    // (C2 bands, ms per step, frame-serial vs 4 lanes: 1/4 frame 0.56 vs 0.63 at k = 2; 1/8 frame 0.375 vs 0.314 at
    // k = 2, 0.295 vs 0.312 at k = 3, 0.273 vs 0.311 at k = 4)
    if (c->opt_in_flight >= 2 ? groups * (uint64_t)c->opt_in_flight >= 2ull * kResidentGroups : groups > 4ull * kResidentGroups) return 1u;
    return spp >= 4 ? 4u : 2u;
}

// Workgroup slots of a launch over tiles_x x tiles_y tiles (trg_kernels.hip block_tile): one per tile in the column order; in the
// XCD-aware order 8 x the largest of the xcd_cols x (8 / xcd_cols) regions (regions differ by a row or a column of tiles at most).
static uint64_t tile_slots(uint32_t tiles_x, uint32_t tiles_y, uint32_t xcd_cols) {
    if (xcd_cols == 0u) return (uint64_t)tiles_x * tiles_y;
    if (xcd_cols == kXcdBlocks) return (uint64_t)((tiles_x + 7u) / 8u) * ((tiles_y + 7u) / 8u) * 64ull;
    if (xcd_cols >= kXcdStripes) {
        const uint32_t S = xcd_cols - kXcdStripes;
        return (uint64_t)((tiles_y + kXcds * S - 1u) / (kXcds * S)) * S * kXcds * tiles_x;
    }
    const uint32_t rc = xcd_cols, rq = kXcds / rc;
    uint64_t largest = 0;
    for (uint32_t ci = 0; ci < rc; ++ci)
        for (uint32_t qi = 0; qi < rq; ++qi) {
            const uint64_t W = (uint64_t)(ci + 1u) * tiles_x / rc - (uint64_t)ci * tiles_x / rc, H = (uint64_t)(qi + 1u) * tiles_y / rq - (uint64_t)qi * tiles_y / rq;
            largest = std::max(largest, W * H);
        }
    return largest * kXcds;
}
// TRG_OPT_TILE_ORDER -1 = the column order for every scene.  Measured on C4 (1,022,244 triangles, regeneration kernel, one launch alone;
// profiles/r03/c4_tile_order.md): the XCD-aware order with 2 x 4 regions cuts the memory-side reads from 50.4 to 32.0 GB per launch, raises the
// L2 hit rate from 81 to 86 % and shortens the mean L1 -> L2 round trip from 245 to 199 cycles -- and takes 22.1 ms instead of 19.3: the
// wavefronts' total lifetime (SQ_WAVE_CYCLES) stays the same, the kernel issues instructions most of the time and what it waits less for
// memory it waits more for an issue slot, while the eight regions differ in cost and the dispatcher deals workgroups to the XCDs
// strictly in turn, so the cheap regions' XCDs idle (1 x 8: 23.8 ms, 4 x 2: 37.5, 8 x 1: 35.9 -- the 16:9 side bars).  TRG_XCD_AUTO_COLS
// (a build flag) would make an XCD-aware layout the automatic choice for scenes traversed from HBM.
static uint32_t choose_xcd_cols(const trg_ctx *c, bool lds_scene, uint32_t tiles_x, uint32_t tiles_y) {
    if (c->opt_tile_order >= (int)kXcdPersist) return (uint32_t)c->opt_tile_order - kXcdPersist;   // 64 + n: the n x (8 / n) regions, dealt through per-XCD queues where the kernel can (regeneration)
    if (c->opt_tile_order >= 0) return (uint32_t)c->opt_tile_order;
    if (lds_scene || TRG_XCD_AUTO_COLS == 0) return 0u;
    uint32_t rc = TRG_XCD_AUTO_COLS;
    while (rc < kXcds && tiles_y < kXcds / rc) rc *= 2u;   // fewer row bands for a launch of few tile rows (a row band of a multi-GPU job)
    while (rc > 1u && tiles_x < rc) rc /= 2u;
    return rc;
}

// global scratch for the stack levels that do not fit in LDS: overflow_levels x grid_threads ints
static int ensure_stack_scratch(trg_ctx *c, const LdsPlan &plan, uint64_t grid_threads, StackDesc &out, int slot = 0) {
    out.klds = plan.klds;
    out.overflow = nullptr;
    if (plan.overflow_levels == 0) return TRG_OK;
    const size_t need = (size_t)plan.overflow_levels * grid_threads * sizeof(int);
    if (need > c->stack_scratch_bytes[slot]) {
        if (c->stack_scratch[slot]) { (void)hipDeviceSynchronize(); (void)hipFree(c->stack_scratch[slot]); c->stack_scratch[slot] = nullptr; c->stack_scratch_bytes[slot] = 0; }
        hipError_t e = hipMalloc((void **)&c->stack_scratch[slot], need);
        if (e != hipSuccess) return fail(c, TRG_ERR_NOMEM, "stack scratch hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        c->stack_scratch_bytes[slot] = need;
    }
    out.overflow = c->stack_scratch[slot];
    return TRG_OK;
}

#if TRG_EXPERIMENTS
static int render_wavefront(trg_ctx *c, uint32_t frame_begin, uint32_t spp, uint32_t bounces, uint32_t row0, uint32_t rows, int slot);
#endif

// Per-launch scratch (traversal-stack overflow, wavefront buffers) is keyed on the STREAM a launch goes to: launches on one stream
// are ordered and may share a buffer, launches on different streams may overlap and never do -- whatever order the caller uses
// its streams in (it used to be launch count modulo the announced number in flight, which was only safe for strict round-robin).
static int scratch_slot(trg_ctx *c) {
    if (c->stream == c->own_stream) return 0;
    for (int k = 1; k < c->slots_used; ++k)
        if (c->slot_stream[k] == c->stream) return k;
    if (c->slots_used >= trg_ctx::kScratchSlots) return -1;
    c->slot_stream[c->slots_used] = c->stream;
    return c->slots_used++;
}

// trg_load_scene with TRG_OPT_GPU_BUILD: LBVH on the device (trg_build.hip), 4-wide nodes only (never LDS-staged).
static int load_scene_gpu_build(trg_ctx *c, const float *pos, const float *nrm, const float *col, const uint32_t *idx,
                                const uint32_t *mat, uint32_t n_verts, uint32_t n_tris) {
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i) {
        const float *p = pos + (size_t)idx[i] * 3;
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    const float diag = std::max({ hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] });
    float maxabs = 0.f;
    for (int a = 0; a < 3; ++a) maxabs = std::max({ maxabs, std::fabs(lo[a]), std::fabs(hi[a]) });
    const float pad = std::max(2e-5f * std::max(diag, 1e-3f), 4e-6f * maxabs);  // same conservative padding as bvh_build.cpp
    struct Dev { void *p = nullptr; ~Dev() { if (p) (void)hipFree(p); } } d_pos, d_idx, d_mat, d_nodes4, d_tris, d_px, d_py, d_quad;
    // quads (bvh_build.h): the pairing is one pass over the index buffer on the host; the builders then work on primitives
    std::vector<uint32_t> px, py;
    const uint32_t n_quads = pair_quads(pos, idx, mat, n_tris, px, py);
    const uint32_t n_prims = (uint32_t)px.size();
    const bool paired = n_quads > 0 && n_prims >= 2;
    HIPCHK(c, hipMalloc(&d_pos.p, (size_t)n_verts * 12)); HIPCHK(c, hipMalloc(&d_idx.p, (size_t)n_tris * 12));
    HIPCHK(c, hipMalloc(&d_mat.p, (size_t)n_tris * 4)); HIPCHK(c, hipMalloc(&d_nodes4.p, (size_t)(n_tris - 1) * 128));
    HIPCHK(c, hipMalloc(&d_tris.p, (size_t)n_tris * 48));
    HIPCHK(c, hipMemcpyAsync(d_pos.p, pos, (size_t)n_verts * 12, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_idx.p, idx, (size_t)n_tris * 12, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_mat.p, mat, (size_t)n_tris * 4, hipMemcpyHostToDevice, c->stream));
    if (paired) {
        HIPCHK(c, hipMalloc(&d_px.p, (size_t)n_prims * 4)); HIPCHK(c, hipMalloc(&d_py.p, (size_t)n_prims * 4)); HIPCHK(c, hipMalloc(&d_quad.p, (size_t)n_tris));
        HIPCHK(c, hipMemcpyAsync(d_px.p, px.data(), (size_t)n_prims * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_py.p, py.data(), (size_t)n_prims * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(d_quad.p, 0, (size_t)n_tris, c->stream));
    }
    uint32_t n4 = 0, depth4 = 0;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    hipError_t e = gpu_build_lbvh((const float *)d_pos.p, (const uint32_t *)d_idx.p, (const uint32_t *)d_mat.p, n_tris, lo, hi, pad,
                                  (float4 *)d_nodes4.p, (float4 *)d_tris.p, &n4, &depth4, c->stream, c->opt_gpu_build == 2 ? 0 : c->opt_gpu_build == 3 ? 2 : 1,
                                  paired ? (const uint32_t *)d_px.p : nullptr, paired ? (const uint32_t *)d_py.p : nullptr, n_prims, (unsigned char *)d_quad.p);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "GPU BVH build failed: %s", hipGetErrorString(e));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_build_ms = ms;
    c->gpu_built = true;
    for (int a = 0; a < 3; ++a) { c->scene_lo[a] = lo[a]; c->scene_hi[a] = hi[a]; }

    SceneDesc sc{};
    sc.n_nodes = 0; sc.n_tris = n_tris;
    sc.n_nodes4 = n4;
    for (int a = 0; a < 3; ++a) sc.center[a] = 0.5f * (lo[a] + hi[a]);
    uint64_t total = 0;
    if (!plan_scene_layout(0, 64u, 0, 0, false, n4, n_tris, sc, total))
        return fail(c, TRG_ERR_RANGE, "trg_load_scene: scene needs %llu B on the device (limit 4 GiB)", (unsigned long long)total);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->blob) { (void)hipFree(c->blob); c->blob = nullptr; c->scene_loaded = false; }
    e = hipMalloc((void **)&c->blob, sc.blob_bytes);
    if (e != hipSuccess) return fail(c, TRG_ERR_NOMEM, "trg_load_scene: hipMalloc(%u) failed: %s", sc.blob_bytes, hipGetErrorString(e));
    HIPCHK(c, hipMemsetAsync(c->blob, 0, sc.blob_bytes, c->stream));
    {   // the attributes in original order, then one pass that writes the 128-byte leaf records (geometry of record i + attributes of its triangle)
        Dev d_nrm, d_col;
        HIPCHK(c, hipMalloc(&d_nrm.p, (size_t)n_tris * 36)); HIPCHK(c, hipMalloc(&d_col.p, (size_t)n_tris * 36));
        HIPCHK(c, hipMemcpyAsync(d_nrm.p, nrm, (size_t)n_tris * 36, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_col.p, col, (size_t)n_tris * 36, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, gpu_fatten_records((const float4 *)d_tris.p, (const float *)d_nrm.p, (const float *)d_col.p, n_tris, c->blob + sc.off_fat, false, sc.center, c->stream));
        HIPCHK(c, gpu_fatten_records((const float4 *)d_tris.p, (const float *)d_nrm.p, (const float *)d_col.p, n_tris, c->blob + sc.off_fat_planes, true, sc.center, c->stream,
                                     (const unsigned char *)d_quad.p));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, gpu_quantize_nodes4((const float4 *)d_nodes4.p, n4, c->blob + sc.off_nodes4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sc.blob = c->blob;
    c->sc = sc;
    c->bvh_nodes = 0; c->bvh_depth = 2 * depth4; c->bvh_leaves = n_tris; c->bvh_quads = 0; c->bvh_boxes = 0;   // (device builds: the quads are paired on the device, not counted)
    c->bvh_nodes4 = n4; c->bvh_depth4 = depth4;
    c->scene_loaded = true;
    LdsPlan plan;
    return plan_lds(c, plan, c->opt_kernel == TRG_KERNEL_POOL);
}

#if TRG_EXPERIMENTS
#include "../../experiments/trg_capi_wavefront.inc.h"   // render_wavefront (TRG_KERNEL_WAVEFRONT)
#endif

namespace trg {
struct HostScene {
    std::vector<unsigned char> blob;   // the image of the device allocation (plan_scene_layout)
    SceneDesc sc{};                    // offsets; sc.blob is filled in per context
    uint32_t bvh_nodes = 0, bvh_depth = 0, bvh_leaves = 0, bvh_nodes4 = 0, bvh_depth4 = 0, bvh_quads = 0, bvh_boxes = 0;
    double build_ms = 0.0;
    float lo[3] = { 0.f, 0.f, 0.f }, hi[3] = { 1.f, 1.f, 1.f };   // bounds of the triangles
};
int ctx_gpu_build_option(const trg_ctx *c) { return c ? c->opt_gpu_build : 0; }
void *ctx_current_stream(trg_ctx *c) { return c ? (void *)c->stream : nullptr; }
int ctx_device(const trg_ctx *c) { return c ? c->device : -1; }
// the per-stream scratch slots (scratch_slot) forget which stream owned them; the buffers stay and go to the next streams that ask.  For a
// caller that has DRAINED the context and is about to destroy its streams (trg_group_set_pipeline): without it every reconfiguration used
// up `depth` more of the 16 slots and the third one at depth 8 failed with "more than 15 different streams".
void ctx_forget_streams(trg_ctx *c) {
    if (!c) return;
    for (int k = 1; k < trg_ctx::kScratchSlots; ++k) c->slot_stream[k] = nullptr;
    c->slots_used = 1;
}
void host_scene_free(HostScene *hs) { delete hs; }
}  // namespace trg

extern "C" {

const char *trg_last_error(trg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
int trg_library_experiments(void) { return TRG_EXPERIMENTS ? 1 : 0; }

int trg_create(trg_ctx **out, int device, uint32_t width, uint32_t height) {
    if (!out || width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(nullptr, TRG_ERR_INVALID, "trg_create: bad arguments (%ux%u)", width, height);
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, TRG_ERR_NODEV, "trg_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, TRG_ERR_INVALID, "trg_create: device %d out of range (%d devices)", device, ndev);
    hipDeviceProp_t prop;
    HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(nullptr, TRG_ERR_NODEV, "trg_create: device %d is %s; kernels are built for gfx950 only", device, prop.gcnArchName);
    HIPCHK(nullptr, hipSetDevice(device));
    trg_ctx *c = new (std::nothrow) trg_ctx;
    if (!c) return fail(nullptr, TRG_ERR_NOMEM, "trg_create: out of host memory");
    c->device = device; c->w = width; c->h = height;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t npix = (size_t)width * height;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipMalloc((void **)&c->accum_own, npix * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&c->offsets, npix * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&c->counters, sizeof(unsigned long long) * kCounterSlots * kCounterWords);
    if (e == hipSuccess) e = hipMalloc((void **)&c->xq, sizeof(uint32_t) * kXcds * trg_ctx::kScratchSlots);
    if (e == hipSuccess) e = hipMemset(c->accum_own, 0, npix * 16);
    if (e == hipSuccess) e = hipMemset(c->counters, 0, sizeof(unsigned long long) * kCounterSlots * kCounterWords);
    if (e != hipSuccess) {
        const int rc = fail(nullptr, e == hipErrorOutOfMemory ? TRG_ERR_NOMEM : TRG_ERR_DEVICE, "trg_create: %s", hipGetErrorString(e));
        trg_destroy(c);
        return rc;
    }
    c->stream = c->own_stream;
    c->accum = c->accum_own;
    *out = c;
    return TRG_OK;
}

void trg_destroy(trg_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->blob) (void)hipFree(c->blob);
    if (c->tex_mem) (void)hipFree(c->tex_mem);
    for (int k = 0; k < trg_ctx::kScratchSlots; ++k) {
        if (c->stack_scratch[k]) (void)hipFree(c->stack_scratch[k]);
        if (c->wf_mem[k]) (void)hipFree(c->wf_mem[k]);
    }
    if (c->counters) (void)hipFree(c->counters);
    if (c->xq) (void)hipFree(c->xq);
    if (c->offsets) (void)hipFree(c->offsets);
    if (c->accum_own) (void)hipFree(c->accum_own);
    for (int k = 0; k < 8; ++k)
        if (c->fence[k]) (void)hipEventDestroy(c->fence[k]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

}  // extern "C" (reopened below)

// argument checks shared by trg_load_scene and host_scene_build
static int check_scene_args(trg_ctx *c, const float *pos, const float *nrm, const float *col, const uint32_t *idx, const uint32_t *mat, uint32_t n_verts, uint32_t n_tris) {
    if (n_tris && (!pos || !nrm || !col || !idx || !mat)) return fail(c, TRG_ERR_INVALID, "trg_load_scene: null buffer");
    if (n_tris >= (1u << 28)) return fail(c, TRG_ERR_RANGE, "trg_load_scene: too many triangles (%u)", n_tris);
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (idx[i] >= n_verts) return fail(c, TRG_ERR_INVALID, "trg_load_scene: index %u out of range (%u vertices) at %zu", idx[i], n_verts, i);
    // refuse a scene that cannot fit the 32-bit blob offsets before anything is built, allocated or copied
    SceneDesc probe{};
    uint64_t least = 0;
    if (!plan_scene_layout(0, 64u, 0, 0, false, 0, n_tris, probe, least))
        return fail(c, TRG_ERR_RANGE, "trg_load_scene: %u triangles need at least %llu B on the device (limit 4 GiB)", n_tris, (unsigned long long)least);
    return TRG_OK;
}

namespace trg {
// Host SAH build + the image of the device blob.  No device work: a device group calls this once and uploads to every context.
int host_scene_build(trg_ctx *c, const float *pos, const float *nrm, const float *col, const uint32_t *idx, const uint32_t *mat, uint32_t n_verts,
                     uint32_t n_tris, HostScene **out) {
    if (!c || !out) return TRG_ERR_INVALID;
    *out = nullptr;
    if (int rc = check_scene_args(c, pos, nrm, col, idx, mat, n_verts, n_tris)) return rc;
    HostScene *hs = new (std::nothrow) HostScene;
    if (!hs) return fail(c, TRG_ERR_NOMEM, "trg_load_scene: out of host memory");
    const auto host_t0 = std::chrono::steady_clock::now();
    Bvh bvh;
    // (boxes: only a scene that may be staged in LDS can use them -- 170 triangles at most; the switch is wider so that the answer does not hinge on it)
    uint32_t box_max_tris = 1024u;
    if (const char *e = getenv("TRG_BVH_BOXES_MAX_TRIS")) box_max_tris = (uint32_t)strtoul(e, nullptr, 10);   // (measurements: boxes as units of the split rule in larger scenes too)
    build_bvh(pos, idx, mat, n_tris, bvh, TRG_WIDE8 != 0, TRG_BOX_LEAVES && !TRG_WIDE8 && (TRG_BOX_LEAVES_HBM || n_tris <= box_max_tris));
    if (TRG_WIDE8 && !bvh.wide8_ok) { delete hs; return fail(c, TRG_ERR_RANGE, "trg_load_scene: the 8-wide layout of this build needs leaves of at most two records (TRG_BVH_MAXLEAF <= 2)"); }
    hs->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    if (n_tris) {
        for (int a = 0; a < 3; ++a) { hs->lo[a] = INFINITY; hs->hi[a] = -INFINITY; }
        for (size_t i = 0; i < (size_t)n_tris * 3; ++i) {
            const float *v = pos + (size_t)idx[i] * 3;
            for (int a = 0; a < 3; ++a) { hs->lo[a] = std::min(hs->lo[a], v[a]); hs->hi[a] = std::max(hs->hi[a], v[a]); }
        }
    }

    // Positions go through the index buffer (MPS vertexBuffer + indexBuffer, MetalRenderer.mm:274-275);
    // normals and colours are read as attributes[triangle*3 + j], NOT through the index buffer
    // (Raytracing.metal:104-108), so they are copied as they are.
    const uint32_t nt_rec = (uint32_t)(bvh.tris.size() / 3);
    const uint32_t attr_tris = std::max(n_tris, 1u);
    // [ LDS nodes | 48-byte triangle records | normals | colours | material ids | Halton tables ] [ quantised 4-wide nodes ] [ 128-byte leaf records ]
    // The first bracket is what a workgroup stages into LDS; only a scene small enough for that has it.  The other two are what
    // the HBM kernels traverse (a small scene kept in HBM -- TRG_OPT_FORCE_GLOBAL, a tree too deep for LDS stacks -- uses them too).
    const uint32_t lds_nodes = bvh.n_nodes;
    const uint64_t small_bytes = (uint64_t)lds_nodes * kLdsNodeBytes + (uint64_t)nt_rec * 50u + (uint64_t)attr_tris * 76u + 80u + kHtabBytes;
    const bool lds_candidate = small_bytes <= kMaxLdsScene && n_tris < (1u << 14);   // (u16 per record: index << 2 | mask)
    SceneDesc sc{};
    sc.n_nodes = lds_candidate ? lds_nodes : 0u; sc.n_tris = n_tris;
    sc.n_tris_rec = lds_candidate ? nt_rec : 0u;
    for (int a = 0; a < 3; ++a) sc.center[a] = n_tris ? 0.5f * (hs->lo[a] + hs->hi[a]) : 0.0f;
    const uint32_t node_bytes = kLdsNodeBytes;
    // a TINY scene (at most kFlatMaxPrims primitives once the quads are paired) also gets the flat list the shipped build tests instead of walking the tree
    uint32_t n_flat = 0;
    if (TRG_EXPERIMENTS && lds_candidate && kFlatMaxPrims && getenv("TRG_FLAT_PRIMS") && atoi(getenv("TRG_FLAT_PRIMS")) != 0) {   // (experiments library, opt-in)
        uint32_t prims = 0;
        for (uint32_t i = 0; i < nt_rec; ++i) { ++prims; if (i < bvh.quad.size() && bvh.quad[i]) ++i; }
        if (n_tris && prims <= kFlatMaxPrims) n_flat = prims;
    }
    sc.n_flat = n_flat;
    sc.n_nodes4 = TRG_WIDE8 ? bvh.n_nodes8 : bvh.n_nodes4;
    const uint32_t n_fat = TRG_WIDE8 ? (uint32_t)bvh.rec8.size() : nt_rec;   // (TRG_WIDE8: two entries per leaf, in the order the 8-wide tree addresses them)
    uint64_t total = 0;
    // the box-leaf flavour of the HBM tree (shipped build): TRG_BVH_BOXES_HBM=0 leaves it out (A/B runs: the kernels then never meet a box code)
    const bool hbm_boxes = TRG_BOX_LEAVES_HBM && !TRG_WIDE8 && bvh.n_nodes4_box != 0 && !(getenv("TRG_BVH_BOXES_HBM") && atoi(getenv("TRG_BVH_BOXES_HBM")) == 0);
    if (!plan_scene_layout(sc.n_nodes, node_bytes, lds_candidate ? nt_rec : 0u, lds_candidate ? attr_tris : 0u, lds_candidate, sc.n_nodes4, n_fat, sc, total,
                           hbm_boxes ? bvh.n_nodes4_box : 0u, hbm_boxes ? bvh.boxes.size() : 0u)) {
        delete hs;
        return fail(c, TRG_ERR_RANGE, "trg_load_scene: scene needs %llu B on the device (limit 4 GiB)", (unsigned long long)total);
    }
    std::vector<unsigned char> &host = hs->blob;
    host.assign(sc.blob_bytes, 0);
    if (sc.n_nodes && lds_candidate) {
        // sign-ordered LDS nodes (trav_node_step_signed): per axis the slab planes of both children as
        // (lo_a, hi_a, lo_b, hi_b) and swapped; children of inner nodes become byte offsets
        for (uint32_t i = 0; i < sc.n_nodes; ++i) {
            const F4 *n = &bvh.nodes[(size_t)i * 4];  // (ax0,ax1,ay0,ay1) (bx0,bx1,by0,by1) (az0,az1,bz0,bz1) (c0,c1,-,-)
            float *o = reinterpret_cast<float *>(&host[sc.off_nodes + (size_t)i * kLdsNodeBytes]);
            const float px[4] = { n[0].x, n[0].y, n[1].x, n[1].y }, py[4] = { n[0].z, n[0].w, n[1].z, n[1].w };
            const float pz[4] = { n[2].x, n[2].y, n[2].z, n[2].w };
            int32_t ch[2];
            memcpy(ch, &n[3].x, 8);
            for (int k = 0; k < 2; ++k) {
                // the root of a box's subtree: ONE leaf of twelve records here (bvh_build.h kLeafBox; its subtree's nodes stay in the array, unreferenced)
                for (const BoxLeaf &bl : bvh.boxes)
                    if (ch[k] >= 0 && (uint32_t)ch[k] == bl.node) ch[k] = ~(int32_t)((bl.first_rec << 3) | kLeafBox);
                if (ch[k] >= 0) ch[k] *= (int32_t)kLdsNodeBytes;   // inner child: byte offset of its node
                // (a leaf child keeps its code ~((record << 3) | (count - 1)): the leaf step turns the record into an address with ONE
                //  v_mad_u32_u24, and the same number indexes the u16 per record of the plane test)
            }
            // one 32-byte block [X copy | Y copy] per sign pair (sx + 2 sy) at 0 / 32 / 64 / 96 -- ONE address for both axes --, then
            // Z+ at 128 and Z- at 160, each followed by the child pair (+ 8 bytes of padding); 192 bytes + 16 of padding
            const float fwd[2][4] = { { px[0], px[1], px[2], px[3] }, { py[0], py[1], py[2], py[3] } };
            const float rev[2][4] = { { px[1], px[0], px[3], px[2] }, { py[1], py[0], py[3], py[2] } };
            for (int sy = 0; sy < 2; ++sy)
                for (int sx = 0; sx < 2; ++sx) {
                    float *q = o + (sx + 2 * sy) * 8;
                    memcpy(q, sx ? rev[0] : fwd[0], 16);
                    memcpy(q + 4, sy ? rev[1] : fwd[1], 16);
                }
            float *zp = o + 32, *zm = o + 40;
            zp[0] = pz[0]; zp[1] = pz[1]; zp[2] = pz[2]; zp[3] = pz[3];
            zm[0] = pz[1]; zm[1] = pz[0]; zm[2] = pz[3]; zm[3] = pz[2];
            memcpy(o + 36, ch, 8);
            memcpy(o + 44, ch, 8);
        }
    }
    if (lds_candidate) {
        memcpy(&host[sc.off_tris], bvh.tris.data(), bvh.tris.size() * sizeof(F4));
        for (uint32_t i = 0; i < nt_rec; ++i)
            fill_plane_record(&host[sc.off_tris_alt + (size_t)i * 48u], reinterpret_cast<uint16_t *>(&host[sc.off_meta]) + i, &bvh.tris[(size_t)i * 3], sc.center,
                              (i < bvh.quad.size() && bvh.quad[i]) ? &bvh.tris[(size_t)(i + 1) * 3] : nullptr);
        // a box leaf's twelve plane records hold six quads X, Y, X, Y ...: a quad's test reads its X record only, so the Y slots are free -- the first
        // takes the box itself, rows (a_k, d_k): l_k = a_k . (P - scene centre) + d_k, the second the X-record offset of each of its six faces
        for (const BoxLeaf &bl : bvh.boxes) {
            if (bl.node == ~0u) continue;   // (a lone quad dressed as a box: the HBM flavour's business)
            float rec[12];
            for (int k = 0; k < 3; ++k) {
                double dk = 0.0;
                for (int a = 0; a < 3; ++a) { rec[k * 4 + a] = bl.axis[k][a]; dk -= (double)bl.axis[k][a] * ((double)bl.center[a] - (double)sc.center[a]); }
                rec[k * 4 + 3] = (float)dk;
            }
            memcpy(&host[sc.off_tris_alt + (size_t)(bl.first_rec + 1u) * 48u], rec, 48);
            uint32_t faces[12] = { 0 };
            for (int f = 0; f < 6; ++f) faces[f] = ((uint32_t)bl.face_rec[f] << 16) | ((uint32_t)bl.face_rec[f] * 48u);
            memcpy(&host[sc.off_tris_alt + (size_t)(bl.first_rec + 3u) * 48u], faces, 48);
        }
        if (sc.n_flat) {   // the flat list: the same plane records, one entry per primitive, + who it is -- in ASCENDING order of original index (traverse_flat's tie rule)
            struct FP { uint32_t key, rec; bool quad; };
            std::vector<FP> fl;
            for (uint32_t i = 0; i < nt_rec; ++i) {
                const bool q = i < bvh.quad.size() && bvh.quad[i];
                uint32_t px, py = ~0u;
                memcpy(&px, &bvh.tris[(size_t)i * 3].w, 4);
                if (q) memcpy(&py, &bvh.tris[(size_t)(i + 1) * 3].w, 4);
                fl.push_back(FP{ std::min(px, py), i, q });
                if (q) ++i;
            }
            std::sort(fl.begin(), fl.end(), [](const FP &x, const FP &y) { return x.key < y.key; });
            for (uint32_t k = 0; k < sc.n_flat && k < fl.size(); ++k) {
                const uint32_t i = fl[k].rec;
                const bool q = fl[k].quad;
                unsigned char *dst = &host[sc.off_flat + (size_t)k * kFlatPrimBytes];
                uint16_t meta;
                fill_plane_record(dst, &meta, &bvh.tris[(size_t)i * 3], sc.center, q ? &bvh.tris[(size_t)(i + 1) * 3] : nullptr);
                uint32_t row3[4] = { 0u, ~0u, 0u, q ? 1u : 0u };
                memcpy(&row3[0], &bvh.tris[(size_t)i * 3].w, 4);
                memcpy(&row3[2], &bvh.tris[(size_t)i * 3 + 1].w, 4);
                if (q) memcpy(&row3[1], &bvh.tris[(size_t)(i + 1) * 3].w, 4);
                memcpy(dst + 48, row3, 16);
            }
        }
        if (n_tris) {
            memcpy(&host[sc.off_normals], nrm, (size_t)n_tris * 36);
            memcpy(&host[sc.off_colors], col, (size_t)n_tris * 36);
            memcpy(&host[sc.off_mats], mat, (size_t)n_tris * 4);
        }
    }
#if TRG_WIDE8
    for (uint32_t e = 0; e < n_fat; ++e) {
        const uint32_t i = bvh.rec8[e], flags = bvh.rec8_flags[e];
        unsigned char *mt = &host[sc.off_fat + (size_t)e * kFatRecBytes], *pl = &host[sc.off_fat_planes + (size_t)e * kFatRecBytes];
        if (i == ~0u) {   // padding behind a one-triangle leaf: mask 0 matches no ray (never addressed anyway: the leaf's flags say so)
            const F4 none[3] = { F4{ 0.f, 0.f, 0.f, 0.f }, F4{ 1.f, 0.f, 0.f, 0.f }, F4{ 0.f, 1.f, 0.f, 0.f } };
            fill_fat_record(mt, none, nrm, col, 0u);
            fill_fat_record_planes(pl, none, nrm, col, 0u, sc.center, nullptr);
            continue;
        }
        fill_fat_record(mt, &bvh.tris[(size_t)i * 3], nrm, col, n_tris);
        fill_fat_record_planes(pl, &bvh.tris[(size_t)i * 3], nrm, col, n_tris, sc.center, (flags & kRec8Quad) ? &bvh.tris[(size_t)(i + 1) * 3] : nullptr);
        // the leaf's flags: float 11 (row 2 .w, unused by Moeller-Trumbore) of the strict build's record; bits 28, 29 of the index word of the shipped one
        memcpy(mt + 44, &flags, 4);
        uint32_t iw;
        memcpy(&iw, pl + 48, 4);
        iw |= flags << 28;
        memcpy(pl + 48, &iw, 4);
    }
#else
    for (uint32_t i = 0; i < nt_rec; ++i) {
        fill_fat_record(&host[sc.off_fat + (size_t)i * kFatRecBytes], &bvh.tris[(size_t)i * 3], nrm, col, n_tris);
        fill_fat_record_planes(&host[sc.off_fat_planes + (size_t)i * kFatRecBytes], &bvh.tris[(size_t)i * 3], nrm, col, n_tris, sc.center,
                               (i < bvh.quad.size() && bvh.quad[i]) ? &bvh.tris[(size_t)(i + 1) * 3] : nullptr);
    }
#endif
    if (lds_candidate) {  // Halton group tables (trg_kernels.h kHtab)
        float *T = reinterpret_cast<float *>(&host[sc.off_htab]);
        for (const HtabSpec &h : kHtab)
            for (uint32_t r = 0; r < h.radix; ++r) {
                double v = 0.0, f = 1.0;
                uint32_t x = r;
                for (uint32_t j = 0; j < h.digits; ++j) { f /= (double)h.base; v += f * (double)(x % h.base); x /= h.base; }
                T[h.offset + r] = (float)v;
            }
    }
#if TRG_WIDE8
    if (sc.n_nodes4) memcpy(&host[sc.off_nodes4], bvh.nodes8.data(), (size_t)sc.n_nodes4 * kQ8NodeBytes);
#else
    if (sc.n_nodes4) memcpy(&host[sc.off_nodes4], bvh.nodes4q.data(), (size_t)sc.n_nodes4 * kQ4NodeBytes);
#endif

    if (hbm_boxes) {
        memcpy(&host[sc.off_nodes4_box], bvh.nodes4q_box.data(), (size_t)bvh.n_nodes4_box * kQ4NodeBytes);
        for (size_t b = 0; b < bvh.boxes.size(); ++b) {
            const BoxLeaf &bl = bvh.boxes[b];
            float rec[16];
            for (int k = 0; k < 3; ++k) {
                double dk = 0.0;
                for (int a = 0; a < 3; ++a) { rec[k * 4 + a] = bl.axis[k][a]; dk -= (double)bl.axis[k][a] * ((double)bl.center[a] - (double)sc.center[a]); }
                rec[k * 4 + 3] = (float)dk;
            }
            uint32_t row3[4] = { bl.first_rec, bl.mask, 0u, 0u };
            for (int f = 0; f < 6; ++f) {
                const uint32_t w7 = (uint32_t)bl.face_rec[f] | ((uint32_t)bl.face_bits[f] << 4);
                row3[2 + f / 4] |= w7 << (7 * (f % 4));
            }
            if (bl.node == ~0u) row3[3] |= 0x80000000u;   // a lone quad dressed as a box: its face is axis 2 (box_hit_resolve)
            memcpy(&rec[12], row3, 16);
            memcpy(&host[sc.off_boxrec + b * (size_t)kFatRecBytes], rec, 64);   // (the second half of the line is padding)
        }
    }

    hs->sc = sc;
    hs->bvh_nodes = bvh.n_nodes; hs->bvh_depth = bvh.depth; hs->bvh_leaves = bvh.n_leaves;
    hs->bvh_quads = bvh.n_quads; hs->bvh_boxes = (lds_candidate || hbm_boxes) ? bvh.n_boxes_real : 0u;
    hs->bvh_nodes4 = TRG_WIDE8 ? bvh.n_nodes8 : bvh.n_nodes4; hs->bvh_depth4 = TRG_WIDE8 ? bvh.depth8 : bvh.depth4;
    *out = hs;
    return TRG_OK;
}

int host_scene_upload(trg_ctx *c, const HostScene *hs) {
    if (!c || !hs) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->tex_mem) { (void)hipFree(c->tex_mem); c->tex_mem = nullptr; c->tex = TexDesc{}; }   // textures belong to the scene they were loaded for
    if (c->blob) { (void)hipFree(c->blob); c->blob = nullptr; c->scene_loaded = false; }
    SceneDesc sc = hs->sc;
    hipError_t e = hipMalloc((void **)&c->blob, sc.blob_bytes);
    if (e != hipSuccess) return fail(c, TRG_ERR_NOMEM, "trg_load_scene: hipMalloc(%u) failed: %s", sc.blob_bytes, hipGetErrorString(e));
    HIPCHK(c, hipMemcpy(c->blob, hs->blob.data(), sc.blob_bytes, hipMemcpyHostToDevice));
    sc.blob = c->blob;
    c->sc = sc;
    c->last_build_ms = hs->build_ms;
    for (int a = 0; a < 3; ++a) { c->scene_lo[a] = hs->lo[a]; c->scene_hi[a] = hs->hi[a]; }
    c->gpu_built = false;
    c->bvh_nodes = hs->bvh_nodes; c->bvh_depth = hs->bvh_depth; c->bvh_leaves = hs->bvh_leaves;
    c->bvh_quads = hs->bvh_quads; c->bvh_boxes = hs->bvh_boxes;
    c->bvh_nodes4 = hs->bvh_nodes4; c->bvh_depth4 = hs->bvh_depth4;
    c->scene_loaded = true;
    LdsPlan plan;
    return plan_lds(c, plan, c->opt_kernel == TRG_KERNEL_POOL);
}
}  // namespace trg

extern "C" {

int trg_load_scene(trg_ctx *c, const float *pos, const float *nrm, const float *col, const uint32_t *idx,
                   const uint32_t *mat, uint32_t n_verts, uint32_t n_tris) {
    if (!c) return TRG_ERR_INVALID;
    if (c->opt_gpu_build && n_tris >= 2 && !TRG_WIDE8) {   // (a TRG_WIDE8 build has the host builder's 8-wide layout only)
        if (int rc = check_scene_args(c, pos, nrm, col, idx, mat, n_verts, n_tris)) return rc;
        HIPCHK(c, hipSetDevice(c->device));
        if (c->tex_mem) {   // textures belong to the scene they were loaded for
            HIPCHK(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->tex_mem); c->tex_mem = nullptr; c->tex = TexDesc{};
        }
        return load_scene_gpu_build(c, pos, nrm, col, idx, mat, n_verts, n_tris);
    }
    HostScene *hs = nullptr;
    if (int rc = host_scene_build(c, pos, nrm, col, idx, mat, n_verts, n_tris, &hs)) return rc;
    const int rc = host_scene_upload(c, hs);
    host_scene_free(hs);
    return rc;
}

int trg_load_textures(trg_ctx *c, const float *uv2, const uint32_t *texture_ids, uint32_t n_tris, const uint8_t *const *images, const uint32_t *widths,
                      const uint32_t *heights, uint32_t n_textures) {
    if (!c) return TRG_ERR_INVALID;
    if (!c->scene_loaded) return fail(c, TRG_ERR_INVALID, "trg_load_textures: load the scene first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->tex_mem) { (void)hipFree(c->tex_mem); c->tex_mem = nullptr; }
    c->tex = TexDesc{};
    if (n_textures == 0) return TRG_OK;
    if (!uv2 || !texture_ids || !images || !widths || !heights) return fail(c, TRG_ERR_INVALID, "trg_load_textures: null argument");
    if (n_tris != c->sc.n_tris) return fail(c, TRG_ERR_INVALID, "trg_load_textures: %u triangles, the scene has %u", n_tris, c->sc.n_tris);
    uint64_t texels = 0;
    for (uint32_t k = 0; k < n_textures; ++k) {
        if (!images[k] || widths[k] == 0 || heights[k] == 0 || widths[k] > 32768u || heights[k] > 32768u)
            return fail(c, TRG_ERR_INVALID, "trg_load_textures: texture %u is empty or larger than 32768", k);
        texels += (uint64_t)widths[k] * heights[k];
    }
    if (texels > 0xFFFFFFF0ull / 4u) return fail(c, TRG_ERR_RANGE, "trg_load_textures: %llu texels (limit 2^30)", (unsigned long long)texels);
    for (uint32_t t = 0; t < n_tris; ++t)
        if (texture_ids[t] > n_textures) return fail(c, TRG_ERR_INVALID, "trg_load_textures: triangle %u names texture %u of %u", t, texture_ids[t], n_textures);
    const size_t uv_bytes = (size_t)n_tris * 24u, id_bytes = (size_t)n_tris * 4u, tab_bytes = (size_t)n_textures * 16u;
    const size_t off_ids = (uv_bytes + 15u) & ~(size_t)15u, off_tab = (off_ids + id_bytes + 15u) & ~(size_t)15u, off_tex = (off_tab + tab_bytes + 15u) & ~(size_t)15u;
    std::vector<unsigned char> host(off_tex + (size_t)texels * 4u, 0);
    memcpy(&host[0], uv2, uv_bytes);
    memcpy(&host[off_ids], texture_ids, id_bytes);
    uint32_t *tab = reinterpret_cast<uint32_t *>(&host[off_tab]);
    uint64_t first = 0;
    for (uint32_t k = 0; k < n_textures; ++k) {
        tab[k * 4 + 0] = (uint32_t)first; tab[k * 4 + 1] = widths[k]; tab[k * 4 + 2] = heights[k]; tab[k * 4 + 3] = 0u;
        memcpy(&host[off_tex + (size_t)first * 4u], images[k], (size_t)widths[k] * heights[k] * 4u);
        first += (uint64_t)widths[k] * heights[k];
    }
    hipError_t e = hipMalloc((void **)&c->tex_mem, host.size());
    if (e != hipSuccess) return fail(c, TRG_ERR_NOMEM, "trg_load_textures: hipMalloc(%zu) failed: %s", host.size(), hipGetErrorString(e));
    HIPCHK(c, hipMemcpy(c->tex_mem, host.data(), host.size(), hipMemcpyHostToDevice));
    c->tex.uv = reinterpret_cast<const float *>(c->tex_mem);
    c->tex.ids = reinterpret_cast<const uint32_t *>(c->tex_mem + off_ids);
    c->tex.table = reinterpret_cast<const uint32_t *>(c->tex_mem + off_tab);
    c->tex.texels = reinterpret_cast<const uint32_t *>(c->tex_mem + off_tex);
    return TRG_OK;
}

int trg_set_uniforms(trg_ctx *c, const trg_uniforms *u) {
    if (!c || !u) return TRG_ERR_INVALID;
    if (u->width != c->w || u->height != c->h)
        return fail(c, TRG_ERR_INVALID, "trg_set_uniforms: uniforms are %ux%u but the context is %ux%u", u->width, u->height, c->w, c->h);
    c->u = *u;
    c->have_uniforms = true;
    return TRG_OK;
}

int trg_set_pixel_offsets(trg_ctx *c, const uint32_t *offsets) {
    if (!c || !offsets) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->offsets, offsets, (size_t)c->w * c->h * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_offsets = true;
    return TRG_OK;
}

int trg_set_pixel_offsets_seed(trg_ctx *c, uint32_t seed) {
    if (!c) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_offsets_fast(seed, c->w * c->h, c->offsets, c->stream));
    c->have_offsets = true;
    return TRG_OK;
}

void trg_microband_rows(uint32_t height, uint32_t n, uint32_t r, uint32_t *rows, uint32_t *stride_rows) {
    const uint32_t nmb = (height + kMicroBandRows - 1u) / kMicroBandRows;          // micro-bands of the image
    const uint32_t mine = (n && r < n && nmb > r) ? (nmb - r + n - 1u) / n : 0u;   // micro-bands r, r + n, ... below nmb
    if (rows) *rows = mine * kMicroBandRows;
    if (stride_rows) *stride_rows = n ? ((nmb + n - 1u) / n) * kMicroBandRows : nmb * kMicroBandRows;
}

// frames [frame_begin, frame_begin + spp) over rows [row0, row0 + rows) of the accumulation buffer: image rows of the same numbers
// (il_n <= 1), or the micro-bands il_r, il_r + il_n, ... of the image stored compactly from row0 on (trg_render_bands)
static int render_impl(trg_ctx *c, uint32_t frame_begin, uint32_t spp, uint32_t bounces, uint32_t row0, uint32_t rows, uint32_t il_n, uint32_t il_r);

int trg_render(trg_ctx *c, uint32_t frame_begin, uint32_t spp, uint32_t bounces, uint32_t row0, uint32_t rows) {
    if (!c) return TRG_ERR_INVALID;
    if (row0 > c->h || rows > c->h - row0) return fail(c, TRG_ERR_INVALID, "trg_render: rows [%u,%u) outside the image (height %u)", row0, row0 + rows, c->h);
    return render_impl(c, frame_begin, spp, bounces, row0, rows, 0u, 0u);
}

int trg_render_bands(trg_ctx *c, uint32_t frame_begin, uint32_t spp, uint32_t bounces, uint32_t n_ranks, uint32_t rank, uint32_t store_row0) {
    if (!c) return TRG_ERR_INVALID;
    if (n_ranks == 0 || rank >= n_ranks) return fail(c, TRG_ERR_INVALID, "trg_render_bands: rank %u of %u", rank, n_ranks);
    uint32_t rows = 0, stride = 0;
    trg_microband_rows(c->h, n_ranks, rank, &rows, &stride);
    // the compact band must fit the buffer it goes to: the context's own one has the image's rows; a caller-owned one (trg_bind_accum) is the
    // caller's to size -- n_ranks * stride rows hold every rank's band
    if (c->accum == c->accum_own && (uint64_t)store_row0 + rows > c->h)
        return fail(c, TRG_ERR_INVALID, "trg_render_bands: rows [%u,%u) do not fit the context's own accumulation buffer (height %u): bind one of n * stride rows", store_row0, store_row0 + rows, c->h);
    return render_impl(c, frame_begin, spp, bounces, store_row0, rows, n_ranks, rank);
}

int trg_unpack_bands(trg_ctx *c, const void *compact, void *image, uint32_t n_ranks) {
    if (!c || !compact || !image || n_ranks == 0 || compact == image) return TRG_ERR_INVALID;
    if ((((uintptr_t)compact) | ((uintptr_t)image)) & 15u) return fail(c, TRG_ERR_INVALID, "trg_unpack_bands: pointers must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t stride = 0;
    trg_microband_rows(c->h, n_ranks, 0, nullptr, &stride);
    const hipError_t e = launch_unpack_bands_fast(static_cast<const float *>(compact), static_cast<float *>(image), c->w, c->h, n_ranks, stride, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_unpack_bands: launch failed: %s", hipGetErrorString(e));
    return TRG_OK;
}

static int render_impl(trg_ctx *c, uint32_t frame_begin, uint32_t spp, uint32_t bounces, uint32_t row0, uint32_t rows, uint32_t il_n, uint32_t il_r) {
    if (!c->scene_loaded) return fail(c, TRG_ERR_INVALID, "trg_render: no scene loaded");
    if (!c->have_uniforms) return fail(c, TRG_ERR_INVALID, "trg_render: uniforms not set");
    if (!c->have_offsets) return fail(c, TRG_ERR_INVALID, "trg_render: pixel offsets not set");
    if (bounces > TRG_MAX_BOUNCES) return fail(c, TRG_ERR_RANGE, "trg_render: %u bounces > TRG_MAX_BOUNCES (%u)", bounces, TRG_MAX_BOUNCES);
    if ((uint64_t)frame_begin + spp > 0xFFFFFFFFull) return fail(c, TRG_ERR_INVALID, "trg_render: frame range overflows");
    if (spp == 0 || rows == 0) { c->last_ms = 0.0; return TRG_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    const int slot = scratch_slot(c);
    if (slot < 0) return fail(c, TRG_ERR_RANGE, "trg_render: more than %d different streams used with this context", trg_ctx::kScratchSlots - 1);
    int kernel = c->opt_kernel;
    if (kernel == TRG_KERNEL_AUTO) kernel = TRG_KERNEL_DIRECT;
    if (il_n > 1u && kernel != TRG_KERNEL_DIRECT)
        return fail(c, TRG_ERR_INVALID, "trg_render_bands: interleaved bands are rendered by the direct megakernel only (TRG_OPT_KERNEL %d)", kernel);
#if TRG_EXPERIMENTS
    if (kernel == TRG_KERNEL_WAVEFRONT) {
        if (c->opt_timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        if (int rc = render_wavefront(c, frame_begin, spp, bounces, row0, rows, slot)) return rc;
        c->renders++; c->launches++; c->last_fsplit = 1; c->last_kernel = TRG_KERNEL_WAVEFRONT;
        if (c->opt_timing) {
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            c->last_ms = ms; c->total_ms += ms;
        }
        return TRG_OK;
    }
    const bool pool = kernel == TRG_KERNEL_POOL;
#else
    const bool pool = false;
#endif
    c->last_kernel = (uint32_t)kernel;
    // frame lanes per workgroup (render_fp_kernel) -- see choose_fsplit
    uint32_t fsplit = pool ? 1u : choose_fsplit(c, spp, rows);
    // a large scene in HBM: the path-regeneration kernel takes the frame lanes itself (so many workgroups per tile)
    const bool regen_wanted = c->opt_regen > 0 || (c->opt_regen < 0 && c->sc.n_tris >= kRegenAutoMinTris);
    uint32_t regen_lanes = 0u;
    if (regen_wanted && !pool) {
        LdsPlan rp;
        if (plan_lds_quiet(c, rp) && !rp.lds_scene && rp.acc_off != 0u) { regen_lanes = fsplit; fsplit = 1u; }
    }
    uint32_t fp_rounds = fsplit > 1 ? std::min<uint32_t>((spp + fsplit - 1) / fsplit, kFpMaxRounds) : 0u;
    if (fp_rounds > kFpMaxRoundsHbm) {   // a scene in HBM also keeps 12 KB of stack levels and 7 KB of parked path state in LDS: two rounds leave room for 6 workgroups per CU
        LdsPlan probe;
        if (plan_lds_quiet(c, probe) && !probe.lds_scene) fp_rounds = kFpMaxRoundsHbm;
    }
    LdsPlan plan;
    if (fsplit > 1 && c->opt_fsplit == 0 && plan_lds(c, plan, false, fp_rounds) != TRG_OK) { fsplit = 1u; fp_rounds = 0u; }  // auto: no room to park
    if (int rc = plan_lds(c, plan, pool, fp_rounds)) return rc;

    RenderParams p{};
    p.u = c->u;
    p.sc = c->sc;
    if (pool && plan.lds_scene) p.sc.lds_stage_bytes = c->sc.off_htab;
    p.offsets = c->offsets;
    p.accum = c->accum;
    p.counters = c->counters;
    p.frame_begin = frame_begin; p.spp = spp; p.bounces = bounces; p.row0 = row0; p.rows = rows;
    p.stack_off = plan.stack_off; p.red_off = plan.red_off; p.pool_off = plan.pool_off;
    p.fsplit = fsplit; p.fp_rounds = fp_rounds; p.acc_off = plan.acc_off;
    p.il_n = il_n > 1u ? il_n : 0u; p.il_r = il_n > 1u ? il_r : 0u;
    p.tex = c->tex;
    // workgroup tile: 16x16 pixels, or (4/fsplit) 8x8 sub-tiles side by side when the frames are split over waves
    const uint32_t tile_w = fsplit > 1 ? 8u * (kWaves / fsplit) : (uint32_t)kTileW, tile_h = fsplit > 1 ? 8u : (uint32_t)kTileH;
    p.tiles_x = (c->w + tile_w - 1) / tile_w;
    const uint32_t tiles_y = (rows + tile_h - 1) / tile_h;
    p.tiles_y = tiles_y;
    p.xcd_cols = choose_xcd_cols(c, plan.lds_scene, p.tiles_x, tiles_y);
    c->last_xcd_cols = p.xcd_cols;
    const uint64_t slots = tile_slots(p.tiles_x, tiles_y, p.xcd_cols);
    if (slots * std::max(regen_lanes, 1u) > 0x7FFFFFFFull) return fail(c, TRG_ERR_RANGE, "trg_render: grid too large");
    const uint32_t grid = (uint32_t)slots;   // workgroup slots (the XCD-aware order pads the launch to 8 equal regions)
    if (int rc = ensure_stack_scratch(c, plan, (uint64_t)grid * std::max(regen_lanes, 1u) * kBlock, p.stack, slot)) return rc;

    // ---- tail compaction (trg_tail.inc.h): scene in LDS, frame-serial direct kernel, enough bounces for paths to die.  Measured on
    //      the Cornell box at 1080p, 16 spp, K = 2 and a re-compaction every second bounce, four frames in flight: 3 bounces 1.862 vs
    //      1.862 ms (off), 4: 2.43 vs 2.57, 5: 2.89 vs 3.24, 6: 3.30 vs 3.90, 8: 4.0 vs 5.3; C3 (64 spp) 15.7 vs 20.7 ms
    uint32_t tail_k = 0;
    if (!pool && fsplit == 1 && plan.lds_scene) {
        if (c->opt_tail > 0) tail_k = (uint32_t)c->opt_tail;
        else if (c->opt_tail < 0 && bounces >= kTailAutoMinBounces) tail_k = kTailAutoK;
        if (tail_k >= bounces) tail_k = 0;
    }
    if (tail_k) {
        const uint32_t fc = std::min<uint32_t>(spp, kTailChunkFrames);
        const uint64_t n_waves = (uint64_t)grid * kWaves, cap = 64ull * fc, band_pixels = (uint64_t)c->w * rows;
        const size_t q_bytes = (size_t)(n_waves * cap * 48u), cnt_bytes = (size_t)((n_waves * 4u + 255u) & ~255ull), rad_bytes = (size_t)(band_pixels * fc * 16u);
        // compaction levels: K, K + 2, K + 4 ... while at least two bounces are left (each level halves the live lanes again)
        // (TRG_OPT_TAIL_REFILL: one level per bounce -- every path of a launch is at the same bounce, its lanes refill from the queue)
        const bool refill = TRG_EXPERIMENTS && (c->opt_tail_refill > 0 || (c->opt_tail_refill < 0 && TRG_TAIL_REFILL_AUTO));
        uint32_t levels[TRG_MAX_BOUNCES + 1]; int n_levels = 0;
        for (uint32_t k = tail_k; k < bounces && n_levels < (int)TRG_MAX_BOUNCES; k += refill ? 1u : kTailLevelStep) { levels[n_levels++] = k; if (c->opt_tail_levels == 1 && !refill) break; }
        const bool sorted = !refill && c->opt_tail_sort > 0 && cap * kWaves <= 4096u;   // the sort kernel holds a tile's keys in LDS
        const bool two_queues = n_levels > 1 || sorted;
        const size_t need = q_bytes * (two_queues ? 2u : 1u) + cnt_bytes * 2u + rad_bytes;
        if (need > c->wf_bytes[slot]) {
            if (c->wf_mem[slot]) { (void)hipDeviceSynchronize(); (void)hipFree(c->wf_mem[slot]); c->wf_mem[slot] = nullptr; c->wf_bytes[slot] = 0; }
            hipError_t me = hipMalloc((void **)&c->wf_mem[slot], need);
            if (me != hipSuccess) return fail(c, TRG_ERR_NOMEM, "tail buffers hipMalloc(%zu) failed: %s", need, hipGetErrorString(me));
            c->wf_bytes[slot] = need;
        }
        p.tail_cap = (uint32_t)cap; p.tail_band_pixels = (uint32_t)band_pixels;
        unsigned char *mem = c->wf_mem[slot];
        void *q[2] = { mem, two_queues ? mem + q_bytes : mem };
        unsigned char *after_q = mem + q_bytes * (two_queues ? 2u : 1u);
        uint32_t *qc[2] = { reinterpret_cast<uint32_t *>(after_q), reinterpret_cast<uint32_t *>(after_q + cnt_bytes) };
        p.tail_radbuf = after_q + 2u * cnt_bytes;
        if (c->opt_timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        for (uint32_t f0 = 0; f0 < spp; f0 += fc) {
            p.frame_begin = frame_begin + f0; p.spp = std::min(fc, spp - f0);
            p.tail_k = levels[0]; p.tail_k_end = levels[0]; p.tail_queue = q[0]; p.tail_count = qc[0];
            hipError_t te = c->opt_strict ? launch_render_head_strict(p, c->opt_counters, grid, plan.total, c->stream) : launch_render_head_fast(p, c->opt_counters, grid, plan.total, c->stream);
            float inv3[3];
            for (int a = 0; a < 3; ++a) inv3[a] = 1.0f / std::max(c->scene_hi[a] - c->scene_lo[a], 1e-20f);
            for (int l = 0; l < n_levels && te == hipSuccess; ++l) {
                p.tail_k = levels[l]; p.tail_k_end = l + 1 < n_levels ? levels[l + 1] : bounces;
                if (sorted) {
                    // TRG_OPT_TAIL_SORT: the paths of a tile by direction octant (and origin cell) before a tail launch reads them -- q0 (written by
                    // the head / the previous tail) -> q1 (read by this tail, which compacts into q0 again); the counts stay in qc[l & 1]
                    p.tail_queue = q[0]; p.tail_count = qc[l & 1]; p.tail_queue_out = q[1];
                    te = c->opt_strict ? launch_tail_sort_strict(p, grid, (uint32_t)c->opt_tail_sort, c->scene_lo, inv3, c->stream)
                                       : launch_tail_sort_fast(p, grid, (uint32_t)c->opt_tail_sort, c->scene_lo, inv3, c->stream);
                    if (te != hipSuccess) break;
                    p.tail_queue = q[1]; p.tail_queue_out = q[0]; p.tail_count_out = qc[(l + 1) & 1];
                } else {
                    p.tail_queue = q[l & 1]; p.tail_count = qc[l & 1]; p.tail_queue_out = q[(l + 1) & 1]; p.tail_count_out = qc[(l + 1) & 1];
                }
#if TRG_EXPERIMENTS
                if (refill) te = c->opt_strict ? launch_render_rtail_strict(p, c->opt_counters, grid, plan.total, c->stream) : launch_render_rtail_fast(p, c->opt_counters, grid, plan.total, c->stream);
                else
#endif
                te = c->opt_strict ? launch_render_tail_strict(p, c->opt_counters, grid, plan.total, c->stream) : launch_render_tail_fast(p, c->opt_counters, grid, plan.total, c->stream);
            }
            if (te == hipSuccess) te = c->opt_strict ? launch_tail_accumulate_strict(p, c->stream) : launch_tail_accumulate_fast(p, c->stream);
            if (te != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_render: tail-compaction launch failed: %s", hipGetErrorString(te));
        }
        c->renders++; c->launches++; c->last_fsplit = 1; c->last_tail_k = tail_k;
        if (c->opt_timing) {
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            c->last_ms = ms; c->total_ms += ms;
        }
        return TRG_OK;
    }
    c->last_tail_k = 0;

    // ---- path regeneration (trg_regen.inc.h): scene in HBM, frame-serial direct kernel.  The frames go in chunks through a
    //      radiance buffer [frame in chunk][pixel in band] that tail_accumulate_kernel folds in frame order.
    // automatic (regen_wanted above): from kRegenAutoMinTris triangles on.  Below, a shading event costs about as much as the two rays it
    // produces and running it for a quarter of a wavefront at a time loses (1080p, 16 spp, 3 bounces, lock step / regeneration in ms: 36
    // triangles 4.2 / 6.0, 2.6 K 8.1 / 9.1, 8.8 K 12.5 / 12.2, 21 K 12.9 / 13.0, 49 K 15.5 / 14.4, 96 K 17.8 / 15.5, 263 K 21.4 / 17.8,
    // 560 K 24.0 / 18.5, 1.02 M 25.9 / 19.4: scripts/regen_crossover.py)
    const bool regen = regen_lanes != 0u;
    c->last_regen = regen ? 1u : 0u;
    if (regen) {
        const uint32_t fc = std::min<uint32_t>(std::max<uint32_t>(spp, 1u), kTailChunkFrames);
        const uint32_t rgrid = grid * regen_lanes;
        const size_t rad_bytes = (size_t)rgrid * kBlock * ((fc + regen_lanes - 1u) / regen_lanes) * 16u;   // one log of 256 x frames records per workgroup
        p.fsplit = regen_lanes;
        if (rad_bytes > c->wf_bytes[slot]) {
            if (c->wf_mem[slot]) { (void)hipDeviceSynchronize(); (void)hipFree(c->wf_mem[slot]); c->wf_mem[slot] = nullptr; c->wf_bytes[slot] = 0; }
            hipError_t me = hipMalloc((void **)&c->wf_mem[slot], rad_bytes);
            if (me != hipSuccess) return fail(c, TRG_ERR_NOMEM, "radiance buffer hipMalloc(%zu) failed: %s", rad_bytes, hipGetErrorString(me));
            c->wf_bytes[slot] = rad_bytes;
        }
        p.tail_radbuf = c->wf_mem[slot];
        // TRG_OPT_TILE_ORDER 64 + n: about as many PERSISTENT workgroups as the chip holds (7 per CU) pop the jobs of the n x (8 / n) screen regions
        // from one queue per XCD (trg_regen.inc.h): the heads are zeroed in front of every launch, on its stream
        const bool persist = c->opt_tile_order >= (int)kXcdPersist && p.xcd_cols != 0u;
        const uint32_t launch_grid = persist ? std::min<uint32_t>(rgrid, (uint32_t)c->cu_count * 7u) : rgrid;
        if (persist) { p.xq = c->xq + (size_t)slot * kXcds; p.xq_jobs = rgrid / kXcds; c->last_xcd_cols = p.xcd_cols + kXcdPersist; }
        if (c->opt_timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        for (uint32_t f0 = 0; f0 < spp; f0 += fc) {
            p.frame_begin = frame_begin + f0; p.spp = std::min(fc, spp - f0);
            if (persist) HIPCHK(c, hipMemsetAsync(p.xq, 0, sizeof(uint32_t) * kXcds, c->stream));
            hipError_t te = c->opt_strict ? launch_render_regen_strict(p, c->opt_counters, launch_grid, plan.total, c->stream)
                                          : launch_render_regen_fast(p, c->opt_counters, launch_grid, plan.total, c->stream);
            if (te == hipSuccess) te = c->opt_strict ? launch_regen_accumulate_strict(p, grid, c->stream) : launch_regen_accumulate_fast(p, grid, c->stream);
            if (te != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_render: regeneration launch failed: %s", hipGetErrorString(te));
        }
        c->renders++; c->launches++; c->last_fsplit = regen_lanes;
        if (c->opt_timing) {
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev1));
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            c->last_ms = ms; c->total_ms += ms;
        }
        return TRG_OK;
    }

    if (c->opt_timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    hipError_t e;
#if TRG_EXPERIMENTS
    if (pool)
        e = c->opt_strict ? launch_render_pool_strict(p, plan.lds_scene, c->opt_counters, grid, plan.total, c->stream)
                          : launch_render_pool_fast(p, plan.lds_scene, c->opt_counters, grid, plan.total, c->stream);
    else
#endif
    if (fsplit > 1)
        e = c->opt_strict ? launch_render_fp_strict(p, plan.lds_scene, c->opt_counters, grid, plan.total, c->stream)
                          : launch_render_fp_fast(p, plan.lds_scene, c->opt_counters, grid, plan.total, c->stream);
    else
        e = c->opt_strict ? launch_render_strict(p, plan.lds_scene, c->opt_counters, grid, plan.total, c->stream)
                          : launch_render_fast(p, plan.lds_scene, c->opt_counters, grid, plan.total, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_render: launch failed: %s", hipGetErrorString(e));
    c->renders++;
    c->launches++;
    c->last_fsplit = fsplit;
    if (c->opt_timing) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        HIPCHK(c, hipEventSynchronize(c->ev1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->last_ms = ms;
        c->total_ms += ms;
    }
    return TRG_OK;
}

int trg_read_accum(trg_ctx *c, float *rgba) {
    if (!c || !rgba) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(rgba, c->accum, (size_t)c->w * c->h * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

int trg_get_stats(trg_ctx *c, trg_stats *out) {
    if (!c || !out) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long host[kCounterSlots * kCounterWords];
    HIPCHK(c, hipMemcpyAsync(host, c->counters, sizeof(host), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long sum[kCounterWords] = { 0 };
    for (int s = 0; s < kCounterSlots; ++s)
        for (int k = 0; k < kCounterWords; ++k) sum[k] += host[s * kCounterWords + k];
    memset(out, 0, sizeof(*out));
    out->primary_rays = sum[0]; out->bounce_rays = sum[1]; out->shadow_rays = sum[2]; out->shaded_hits = sum[3];
    out->node_fetches = sum[4]; out->tri_tests = sum[5];
    out->wave_node_iters = sum[6]; out->wave_tri_iters = sum[7];
    out->last_render_ms = c->last_ms; out->total_render_ms = c->total_ms; out->renders = c->renders;
    out->bvh_nodes = c->bvh_nodes; out->bvh_depth = c->bvh_depth; out->bvh_leaves = c->bvh_leaves;
    out->bvh_quads = c->bvh_quads; out->bvh_boxes = c->bvh_boxes;
    out->scene_bytes = c->sc.blob_bytes;
    out->last_build_ms = c->last_build_ms; out->gpu_built = c->gpu_built ? 1u : 0u;
    out->bvh_nodes4 = c->bvh_nodes4; out->bvh_depth4 = c->bvh_depth4;
    out->last_frame_split = c->last_fsplit;
    out->last_tail_bounce = c->last_tail_k;
    out->last_kernel = c->last_kernel;
    out->last_regen = c->last_regen;
    out->last_tile_order = c->last_xcd_cols;
    if (c->scene_loaded) {
        LdsPlan plan;
        if (plan_lds(c, plan, c->opt_kernel == TRG_KERNEL_POOL) == TRG_OK) { out->scene_in_lds = plan.lds_scene ? 1u : 0u; out->lds_bytes = plan.total; }
    }
    return TRG_OK;
}

int trg_reset_stats(trg_ctx *c) {
    if (!c) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long) * kCounterSlots * kCounterWords, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->last_ms = 0.0; c->total_ms = 0.0; c->renders = 0;
    return TRG_OK;
}

int trg_set_option(trg_ctx *c, int option, int64_t value) {
    if (!c) return TRG_ERR_INVALID;
    switch (option) {
    case TRG_OPT_STRICT: c->opt_strict = value != 0; break;
    case TRG_OPT_COUNTERS: c->opt_counters = value != 0; break;
    case TRG_OPT_FORCE_GLOBAL: c->opt_force_global = value != 0; break;
    case TRG_OPT_TIMING: c->opt_timing = value != 0; break;
    case TRG_OPT_GPU_BUILD:
        if (value < 0 || value > 3) return fail(c, TRG_ERR_INVALID, "trg_set_option: GPU build must be 0 (host SAH), 1 (device SAH), 2 (device LBVH) or 3 (device PLOC)");
        c->opt_gpu_build = (int)value;
        break;
    case TRG_OPT_FRAME_SPLIT:
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(c, TRG_ERR_INVALID, "trg_set_option: frame split must be 0 (auto), 1, 2 or 4");
        c->opt_fsplit = (int)value;
        break;
    case TRG_OPT_TAIL_BOUNCE:
        if (value < -1 || value > (int64_t)TRG_MAX_BOUNCES) return fail(c, TRG_ERR_INVALID, "trg_set_option: tail bounce must be -1 (auto), 0 (off) or 1..%u", TRG_MAX_BOUNCES);
        c->opt_tail = (int)value;
        break;
    case TRG_OPT_TAIL_LEVELS:
        c->opt_tail_levels = value == 1 ? 1 : 0;
        break;
    case TRG_OPT_REGEN:
        if (value < -1 || value > 1) return fail(c, TRG_ERR_INVALID, "trg_set_option: regeneration must be -1 (auto), 0 (off) or 1 (on)");
        c->opt_regen = (int)value;
        break;
    case TRG_OPT_STACK_LDS_LEVELS:
        if (value < 2 || value > (int64_t)kStackLdsLevels) return fail(c, TRG_ERR_INVALID, "trg_set_option: 2..%u stack levels in LDS", kStackLdsLevels);
        c->opt_stack_levels = (int)value;
        break;
    case TRG_OPT_TAIL_SORT:
        if (value < 0 || value > 3) return fail(c, TRG_ERR_INVALID, "trg_set_option: tail sort must be 0 (off), 1 (octant), 2 or 3 (octant + origin cell)");
        c->opt_tail_sort = (int)value;
        break;
    case TRG_OPT_TAIL_REFILL:
        if (value < -1 || value > 1) return fail(c, TRG_ERR_INVALID, "trg_set_option: tail refill must be -1 (auto), 0 (off) or 1 (on)");
        if (!TRG_EXPERIMENTS && value == 1) return fail(c, TRG_ERR_INVALID, "trg_set_option: the refilling tail kernel is an experiment that lost to the lock-step one and is not in this library: "
                                                                           "experiments/lib/libtoyraygun_hip_exp.so has it (profiles/r05/c3_tail_refill_experiment.md)");
        c->opt_tail_refill = (int)value;
        break;
    case TRG_OPT_TILE_ORDER:
        if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 17 && value != 18 && value != 20 && value != 24 && value != 32 && value != 65 && value != 66 && value != 68 && value != 72)
            return fail(c, TRG_ERR_INVALID, "trg_set_option: tile order must be -1 (auto), 0 (columns), 1, 2, 4, 8 (XCD regions: column strips), 17, 18, 20, 24 (XCD stripes of 1, 2, 4, 8 tile rows), 32 (XCD sub-blocks of 8 x 8-tile super-blocks) or 65, 66, 68, 72 (the XCD regions through per-XCD job queues)");
        c->opt_tile_order = (int)value;
        break;
    case TRG_OPT_LAUNCHES_IN_FLIGHT:
        if (value < 1 || value > 16) return fail(c, TRG_ERR_INVALID, "trg_set_option: launches in flight must be 1..16");
        c->opt_in_flight = (int)value;
        break;
    case TRG_OPT_KERNEL:
        if (value != TRG_KERNEL_DIRECT && value != TRG_KERNEL_POOL && value != TRG_KERNEL_WAVEFRONT && value != TRG_KERNEL_AUTO)
            return fail(c, TRG_ERR_INVALID, "trg_set_option: unknown kernel %lld", (long long)value);
        if (!TRG_EXPERIMENTS && (value == TRG_KERNEL_POOL || value == TRG_KERNEL_WAVEFRONT))
            return fail(c, TRG_ERR_INVALID, "trg_set_option: the %s schedule is an experiment that lost to the direct megakernel and is not in this library: "
                                            "experiments/lib/libtoyraygun_hip_exp.so has it (experiments/README.md)", value == TRG_KERNEL_POOL ? "path-pool" : "wavefront");
        c->opt_kernel = (int)value;
        break;
    default: return fail(c, TRG_ERR_INVALID, "trg_set_option: unknown option %d", option);
    }
    return TRG_OK;
}

int trg_bind_accum(trg_ctx *c, void *device_ptr) {
    if (!c) return TRG_ERR_INVALID;
    if (device_ptr && ((uintptr_t)device_ptr & 15u)) return fail(c, TRG_ERR_INVALID, "trg_bind_accum: pointer must be 16-byte aligned");
    c->accum = device_ptr ? static_cast<float *>(device_ptr) : c->accum_own;
    return TRG_OK;
}

int trg_accum_device_ptr(trg_ctx *c, void **out) {
    if (!c || !out) return TRG_ERR_INVALID;
    *out = c->accum;
    return TRG_OK;
}

int trg_set_stream(trg_ctx *c, void *hip_stream) {
    if (!c) return TRG_ERR_INVALID;
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return TRG_OK;
}

int trg_stream_idle(trg_ctx *c) {
    if (!c) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(c, TRG_ERR_DEVICE, "hipStreamQuery failed: %s", hipGetErrorString(e));
}

int trg_fence_record(trg_ctx *c, int slot) {
    if (!c || slot < 0 || slot >= 8) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->fence[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->fence[slot], hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->fence[slot], c->stream));
    c->fence_set[slot] = true;
    return TRG_OK;
}

int trg_fence_wait(trg_ctx *c, int slot) {
    if (!c || slot < 0 || slot >= 8) return TRG_ERR_INVALID;
    if (!c->fence_set[slot]) return TRG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->fence[slot]));
    return TRG_OK;
}

int trg_sync(trg_ctx *c) {
    if (!c) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

// ---- host-only introspection: build the BVH exactly as trg_load_scene does and hand back the flattened
//      arrays (no GPU needed).  nodes_out: 16 floats per node, tris_out: 12 floats per record; either may be
//      NULL to query sizes only. ----
int trg_debug_build_bvh(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts,
                        uint32_t n_tris, float *nodes_out, uint32_t nodes_cap, float *tris_out, uint32_t tris_cap,
                        uint32_t *n_nodes, uint32_t *n_tri_records, uint32_t *depth) {
    if (n_tris && (!positions3 || !indices || !material_ids)) return TRG_ERR_INVALID;
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_verts) return TRG_ERR_INVALID;
    Bvh bvh;
    build_bvh(positions3, indices, material_ids, n_tris, bvh, false, debug_want_boxes());
    const uint32_t nrec = (uint32_t)(bvh.tris.size() / 3);
    if (n_nodes) *n_nodes = bvh.n_nodes;
    if (n_tri_records) *n_tri_records = nrec;
    if (depth) *depth = bvh.depth;
    if (nodes_out) {
        if (nodes_cap < bvh.n_nodes) return TRG_ERR_RANGE;
        memcpy(nodes_out, bvh.nodes.data(), (size_t)bvh.n_nodes * 64);
    }
    if (tris_out) {
        if (tris_cap < nrec) return TRG_ERR_RANGE;
        memcpy(tris_out, bvh.tris.data(), (size_t)nrec * 48);
    }
    return TRG_OK;
}

int trg_debug_build_bvh4(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts,
                         uint32_t n_tris, float *nodes4_out, uint32_t nodes4_cap, uint32_t *n_nodes4, uint32_t *depth4) {
    if (n_tris && (!positions3 || !indices || !material_ids)) return TRG_ERR_INVALID;
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_verts) return TRG_ERR_INVALID;
    Bvh bvh;
    build_bvh(positions3, indices, material_ids, n_tris, bvh, false, debug_want_boxes());
    if (n_nodes4) *n_nodes4 = bvh.n_nodes4;
    if (depth4) *depth4 = bvh.depth4;
    if (nodes4_out) {
        if (nodes4_cap < bvh.n_nodes4) return TRG_ERR_RANGE;
        memcpy(nodes4_out, bvh.nodes4.data(), (size_t)bvh.n_nodes4 * 128);
    }
    return TRG_OK;
}

int trg_debug_boxes(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris,
                    float *boxes20_out, uint32_t boxes_cap, uint32_t *n_boxes) {
    if (n_tris && (!positions3 || !indices || !material_ids)) return TRG_ERR_INVALID;
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_verts) return TRG_ERR_INVALID;
    Bvh bvh;
    build_bvh(positions3, indices, material_ids, n_tris, bvh, false, true);
    if (n_boxes) *n_boxes = bvh.n_boxes_real;
    if (boxes20_out) {
        if (boxes_cap < bvh.n_boxes_real) return TRG_ERR_RANGE;
        for (size_t b = 0; b < bvh.n_boxes_real; ++b) {
            const BoxLeaf &bl = bvh.boxes[b];
            float *o = boxes20_out + b * 20;
            o[0] = (float)bl.node; o[1] = (float)bl.first_rec;
            for (int a = 0; a < 3; ++a) { o[2 + a] = bl.center[a]; for (int k = 0; k < 3; ++k) o[5 + k * 3 + a] = bl.axis[k][a]; }
            for (int f = 0; f < 6; ++f) o[14 + f] = (float)bl.face_rec[f];
        }
    }
    return TRG_OK;
}

int trg_debug_build_bvh4q(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts,
                          uint32_t n_tris, uint32_t *nodes4q_out, uint32_t nodes4_cap, uint32_t *n_nodes4) {
    if (n_tris && (!positions3 || !indices || !material_ids)) return TRG_ERR_INVALID;
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_verts) return TRG_ERR_INVALID;
    Bvh bvh;
    build_bvh(positions3, indices, material_ids, n_tris, bvh, false, debug_want_boxes());
    if (n_nodes4) *n_nodes4 = bvh.n_nodes4;
    if (nodes4q_out) {
        if (nodes4_cap < bvh.n_nodes4) return TRG_ERR_RANGE;
        memcpy(nodes4q_out, bvh.nodes4q.data(), (size_t)bvh.n_nodes4 * kQ4NodeBytes);
    }
    return TRG_OK;
}

int trg_debug_leaf_records(const float *positions3, const float *normals3, const float *colors3, const uint32_t *indices, const uint32_t *material_ids,
                           uint32_t n_verts, uint32_t n_tris, float *records32_out, uint32_t records_cap, uint32_t *n_records) {
    if (n_tris && (!positions3 || !normals3 || !colors3 || !indices || !material_ids)) return TRG_ERR_INVALID;
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_verts) return TRG_ERR_INVALID;
    Bvh bvh;
    build_bvh(positions3, indices, material_ids, n_tris, bvh, false, debug_want_boxes());
    const uint32_t nrec = (uint32_t)(bvh.tris.size() / 3);
    if (n_records) *n_records = nrec;
    if (records32_out) {
        if (records_cap < nrec) return TRG_ERR_RANGE;
        for (uint32_t i = 0; i < nrec; ++i)
            fill_fat_record(reinterpret_cast<unsigned char *>(records32_out) + (size_t)i * kFatRecBytes, &bvh.tris[(size_t)i * 3], normals3, colors3, n_tris);
    }
    return TRG_OK;
}

int trg_debug_plane_records(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris,
                            float *planes12_out, uint16_t *meta_out, uint32_t records_cap, uint32_t *n_records, float *center3_out) {
    if (!positions3 || !indices || !material_ids || n_tris == 0) return TRG_ERR_INVALID;
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        if (indices[i] >= n_verts) return TRG_ERR_INVALID;
    Bvh bvh;
    build_bvh(positions3, indices, material_ids, n_tris, bvh, false, debug_want_boxes());
    const uint32_t nrec = (uint32_t)(bvh.tris.size() / 3);
    if (n_records) *n_records = nrec;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY }, ctr[3];
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], positions3[(size_t)indices[i] * 3 + a]); hi[a] = std::max(hi[a], positions3[(size_t)indices[i] * 3 + a]); }
    for (int a = 0; a < 3; ++a) ctr[a] = 0.5f * (lo[a] + hi[a]);
    if (center3_out) memcpy(center3_out, ctr, 12);
    if (planes12_out || meta_out) {
        if (records_cap < nrec || !planes12_out || !meta_out) return TRG_ERR_RANGE;
        for (uint32_t i = 0; i < nrec; ++i)
            fill_plane_record(reinterpret_cast<unsigned char *>(planes12_out) + (size_t)i * 48u, meta_out + i, &bvh.tris[(size_t)i * 3], ctr,
                              (i < bvh.quad.size() && bvh.quad[i]) ? &bvh.tris[(size_t)(i + 1) * 3] : nullptr);
    }
    return TRG_OK;
}

int trg_debug_tile_of_slot(uint32_t tiles_x, uint32_t tiles_y, uint32_t order, uint32_t slot, uint32_t *n_slots, uint32_t *bx, uint32_t *by) {
    if (tiles_x == 0 || tiles_y == 0 || (order != 0 && order != 1 && order != 2 && order != 4 && order != 8 && order != 17 && order != 18 && order != 20 && order != 24 && order != 32)) return TRG_ERR_INVALID;
    const uint64_t slots = tile_slots(tiles_x, tiles_y, order);
    if (slots > 0x7FFFFFFFull) return TRG_ERR_RANGE;
    if (n_slots) *n_slots = (uint32_t)slots;
    uint32_t x = 0, y = 0;
    const bool valid = slot < slots && tile_of_slot(tiles_x, tiles_y, order, slot, x, y);
    if (bx) *bx = x;
    if (by) *by = y;
    return valid ? 1 : 0;
}

int trg_debug_xcc_ids(trg_ctx *c, uint32_t n_blocks, uint32_t *out) {
    if (!c || (n_blocks && !out) || n_blocks > (1u << 20)) return TRG_ERR_INVALID;
    if (n_blocks == 0) return TRG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    struct Dev { void *p = nullptr; ~Dev() { if (p) (void)hipFree(p); } } d;
    HIPCHK(c, hipMalloc(&d.p, (size_t)n_blocks * 4));
    HIPCHK(c, launch_xcc_probe_fast(n_blocks, (uint32_t *)d.p, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, d.p, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

int trg_debug_scene_layout(uint64_t n_tris, uint64_t n_nodes4, uint64_t *total_bytes, uint32_t *offsets7) {
    SceneDesc sc{};
    uint64_t total = 0;
    const bool ok = plan_scene_layout(0, 64u, 0, 0, false, n_nodes4, n_tris ? n_tris : 1u, sc, total);
    if (total_bytes) *total_bytes = total;
    if (!ok) return TRG_ERR_RANGE;
    if (offsets7) {
        const uint32_t o[7] = { sc.off_nodes, sc.off_tris, sc.off_normals, sc.off_colors, sc.off_mats, sc.off_nodes4, sc.off_fat };
        memcpy(offsets7, o, sizeof(o));
    }
    return TRG_OK;
}

// ---- stage-level entry points (host buffers in, host buffers out) ----
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
};
}  // namespace

int trg_trace(trg_ctx *c, const trg_ray *rays, size_t n, int any_hit, void *out) {
    if (!c || (n && (!rays || !out))) return TRG_ERR_INVALID;
    if (!c->scene_loaded) return fail(c, TRG_ERR_INVALID, "trg_trace: no scene loaded");
    if (n > 0x7FFFFFFFull) return fail(c, TRG_ERR_RANGE, "trg_trace: too many rays");
    if (n == 0) return TRG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    LdsPlan plan;
    {
        // The shipped build's plane test on an LDS-resident scene sees the two low bits of a material id only (the u16 per record:
        // trg_device.h trav_begin) -- all the renderers' ray masks 3 and 1 can ask for.  Rays that carry other mask bits are traced through the
        // HBM records, which hold the whole material id (MetalRenderer.mm:269,276: ray.mask & triangle mask, 32 bits).
        bool wide_mask = false;
        if (!c->opt_strict && TRG_TRI_PLANES)
            for (size_t k = 0; k < n && !wide_mask; ++k) wide_mask = (rays[k].mask & ~3u) != 0u;
        const bool saved = c->opt_force_global;
        if (wide_mask) c->opt_force_global = true;
        const int rc = plan_lds(c, plan);
        c->opt_force_global = saved;
        if (rc) return rc;
    }
    DevBuf dr, dout;
    const size_t out_bytes = n * (any_hit ? sizeof(float) : sizeof(trg_isect));
    HIPCHK(c, dr.alloc(n * sizeof(trg_ray)));
    HIPCHK(c, dout.alloc(out_bytes));
    HIPCHK(c, hipMemcpyAsync(dr.p, rays, n * sizeof(trg_ray), hipMemcpyHostToDevice, c->stream));
    TraceParams p{};
    p.sc = c->sc; p.rays = static_cast<const trg_ray *>(dr.p); p.out = dout.p; p.n = (uint32_t)n; p.stack_off = plan.stack_off;
    const int slot = scratch_slot(c);
    if (slot < 0) return fail(c, TRG_ERR_RANGE, "trg_trace: more than %d different streams used with this context", trg_ctx::kScratchSlots - 1);
    if (int rc = ensure_stack_scratch(c, plan, (uint64_t)((n + kBlock - 1) / kBlock) * kBlock, p.stack, slot)) return rc;
    hipError_t e = c->opt_strict ? launch_trace_strict(p, plan.lds_scene, any_hit != 0, plan.total, c->stream)
                                 : launch_trace_fast(p, plan.lds_scene, any_hit != 0, plan.total, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_trace: launch failed: %s", hipGetErrorString(e));
    HIPCHK(c, hipMemcpyAsync(out, dout.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

int trg_halton(trg_ctx *c, const uint32_t *i, const uint32_t *d, size_t n, float *out) {
    if (!c || (n && (!i || !d || !out)) || n > 0x7FFFFFFFull) return TRG_ERR_INVALID;
    if (n == 0) return TRG_OK;
    for (size_t k = 0; k < n; ++k)
        if (d[k] >= 64) return fail(c, TRG_ERR_RANGE, "trg_halton: dimension %u >= 64", d[k]);
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf di, dd, dout;
    HIPCHK(c, di.alloc(n * 4)); HIPCHK(c, dd.alloc(n * 4)); HIPCHK(c, dout.alloc(n * 4));
    HIPCHK(c, hipMemcpyAsync(di.p, i, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dd.p, d, n * 4, hipMemcpyHostToDevice, c->stream));
    hipError_t e = c->opt_strict ? launch_halton_strict((const uint32_t *)di.p, (const uint32_t *)dd.p, (uint32_t)n, (float *)dout.p, c->stream)
                                 : launch_halton_fast((const uint32_t *)di.p, (const uint32_t *)dd.p, (uint32_t)n, (float *)dout.p, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_halton: launch failed: %s", hipGetErrorString(e));
    HIPCHK(c, hipMemcpyAsync(out, dout.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

int trg_halton_table(trg_ctx *c, const uint32_t *i, const uint32_t *d, size_t n, float *out) {
    if (!c || (n && (!i || !d || !out)) || n > 0x7FFFFFFFull) return TRG_ERR_INVALID;
    if (!c->scene_loaded || c->sc.lds_stage_bytes == 0)
        return fail(c, TRG_ERR_INVALID, "trg_halton_table: needs a loaded scene small enough for LDS (the group tables are staged with it)");
    if (n == 0) return TRG_OK;
    for (size_t k = 0; k < n; ++k)
        if (d[k] >= 64) return fail(c, TRG_ERR_RANGE, "trg_halton_table: dimension %u >= 64", d[k]);
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf di, dd, dout;
    HIPCHK(c, di.alloc(n * 4)); HIPCHK(c, dd.alloc(n * 4)); HIPCHK(c, dout.alloc(n * 4));
    HIPCHK(c, hipMemcpyAsync(di.p, i, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dd.p, d, n * 4, hipMemcpyHostToDevice, c->stream));
    // always the shipped (fast) build: the strict build never uses the tables
    hipError_t e = launch_halton_tab_fast(reinterpret_cast<const float *>(c->blob + c->sc.off_htab), (const uint32_t *)di.p, (const uint32_t *)dd.p,
                                          (uint32_t)n, (float *)dout.p, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_halton_table: launch failed: %s", hipGetErrorString(e));
    HIPCHK(c, hipMemcpyAsync(out, dout.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

int trg_raygen(trg_ctx *c, uint32_t frameIndex, trg_ray *out) {
    if (!c || !out) return TRG_ERR_INVALID;
    if (!c->have_uniforms || !c->have_offsets) return fail(c, TRG_ERR_INVALID, "trg_raygen: uniforms / pixel offsets not set");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->w * c->h;
    DevBuf dout;
    HIPCHK(c, dout.alloc(n * sizeof(trg_ray)));
    trg_uniforms u = c->u;
    u.frameIndex = frameIndex;
    hipError_t e = c->opt_strict ? launch_raygen_strict(u, c->offsets, (trg_ray *)dout.p, c->stream)
                                 : launch_raygen_fast(u, c->offsets, (trg_ray *)dout.p, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_raygen: launch failed: %s", hipGetErrorString(e));
    HIPCHK(c, hipMemcpyAsync(out, dout.p, n * sizeof(trg_ray), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

int trg_sample(trg_ctx *c, const float *p3, const float *n3, const float *r4, size_t n, float *out12) {
    if (!c || (n && (!p3 || !n3 || !r4 || !out12)) || n > 0x0FFFFFFFull) return TRG_ERR_INVALID;
    if (!c->have_uniforms) return fail(c, TRG_ERR_INVALID, "trg_sample: uniforms not set");
    if (n == 0) return TRG_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf dp, dn, dr, dout;
    HIPCHK(c, dp.alloc(n * 12)); HIPCHK(c, dn.alloc(n * 12)); HIPCHK(c, dr.alloc(n * 16)); HIPCHK(c, dout.alloc(n * 48));
    HIPCHK(c, hipMemcpyAsync(dp.p, p3, n * 12, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dn.p, n3, n * 12, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dr.p, r4, n * 16, hipMemcpyHostToDevice, c->stream));
    hipError_t e = c->opt_strict ? launch_sample_strict(c->u, (const float *)dp.p, (const float *)dn.p, (const float *)dr.p, (uint32_t)n, (float *)dout.p, c->stream)
                                 : launch_sample_fast(c->u, (const float *)dp.p, (const float *)dn.p, (const float *)dr.p, (uint32_t)n, (float *)dout.p, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_sample: launch failed: %s", hipGetErrorString(e));
    HIPCHK(c, hipMemcpyAsync(out12, dout.p, n * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

int trg_postprocess(trg_ctx *c, uint8_t *rgba8, int flip_y) {
    if (!c || !rgba8) return TRG_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->w * c->h;
    DevBuf dout;
    HIPCHK(c, dout.alloc(n * 4));
    hipError_t e = c->opt_strict ? launch_postprocess_strict(c->accum, c->w, c->h, (uint8_t *)dout.p, flip_y, c->stream)
                                 : launch_postprocess_fast(c->accum, c->w, c->h, (uint8_t *)dout.p, flip_y, c->stream);
    if (e != hipSuccess) return fail(c, TRG_ERR_DEVICE, "trg_postprocess: launch failed: %s", hipGetErrorString(e));
    HIPCHK(c, hipMemcpyAsync(rgba8, dout.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TRG_OK;
}

}  // extern "C"
