/*
 * trg.h -- C ABI of the MI355X (gfx950) path-tracing layer: libtoyraygun_hip.so.
 *
 * This is the drop-in boundary of SURVEY.md section 8(b).  The reference (andr3wmac/ToyRaygun)
 * has no C ABI of its own: its backends are C++ subclasses of toyraygun::Renderer that call a
 * closed vendor intersector (MPS / DXR).  Each entry point below names the reference interface
 * it replaces (file:line under the reference tree).  Only plain pointers and sizes cross the
 * ABI; no exception does; every function returns TRG_OK (0) or a negative error code and
 * trg_last_error() gives the message.  One context per device, one host thread per context,
 * all kernels of a context on one HIP stream.
 */
#ifndef TRG_H
#define TRG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: only these entry points are exported */
#define TRG_API __attribute__((visibility("default")))

#define TRG_OK 0
#define TRG_ERR_INVALID (-22)  /* bad argument / call order        */
#define TRG_ERR_NOMEM (-12)    /* device or host allocation failed */
#define TRG_ERR_DEVICE (-5)    /* HIP runtime error                */
#define TRG_ERR_NODEV (-19)    /* no usable gfx950 device          */
#define TRG_ERR_RANGE (-34)    /* scene exceeds a kernel limit (BVH depth, bounce count) */

#define TRG_MATERIAL_DEFAULT 1u  /* src/engine/Renderer.h:16 */
#define TRG_MATERIAL_EMISSIVE 2u /* src/engine/Renderer.h:17 */
#define TRG_MAX_BOUNCES 15u      /* Halton dimension 2 + 4*bounce + 3 must stay < 64 */

typedef struct trg_ctx trg_ctx;

/* The 176-byte shader `Uniforms` block: runtime/shaders/common.h:14-34 as laid out by
 * src/engine/Uniforms.h:19-41 + src/engine/Metal/MetalUniforms.h:16-60.  inv_view_proj is the memory
 * image of the simd float4x4, i.e. element [col*4+row] = bx row-major invViewProj[row*4+col]. */
typedef struct trg_uniforms {
    uint32_t width, height, frameIndex, _pad;
    float cam_pos[4];
    float inv_view_proj[16];
    float light_pos[4], light_forward[4], light_right[4], light_up[4], light_color[4];
} trg_uniforms;

/* runtime/shaders/metal/Raytracing.metal:21-28 (48 B) and :31-38 (16 B) */
typedef struct trg_ray {
    float origin[3];
    uint32_t mask;
    float direction[3];
    float maxDistance;
    float color[4];
} trg_ray;
typedef struct trg_isect {
    float distance;          /* < 0: miss or inactive ray */
    int32_t primitiveIndex;  /* index into the ORIGINAL triangle order; -1 on miss */
    float coordinates[2];    /* barycentric weights of vertex 0 and vertex 1 */
} trg_isect;

typedef struct trg_stats {
    /* ray counts accumulated over every trg_render since create / trg_reset_stats.  A ray is
     * counted iff it is actually traversed: primary rays, bounce rays (nearest-hit, bounce >= 1)
     * and shadow rays (any-hit) submitted with maxDistance >= 0. */
    uint64_t primary_rays, bounce_rays, shadow_rays, shaded_hits;
    /* filled only while TRG_OPT_COUNTERS is on: node steps in units of two child boxes (a BVH2 node of an
     * LDS-resident scene = 1, a 4-wide node of an HBM-resident scene = 2) and ray/triangle tests */
    uint64_t node_fetches, tri_tests;
    /* wave-level loop trips of the traversal (TRG_OPT_COUNTERS): node_fetches / (64 * wave_node_iters) is the
     * lane utilisation of the node loop, likewise for triangles */
    uint64_t wave_node_iters, wave_tri_iters;
    double last_render_ms;  /* HIP-event time of the kernels of the last trg_render, on its stream */
    double total_render_ms; /* sum over renders since reset */
    uint32_t renders;       /* trg_render calls since reset */
    uint32_t bvh_nodes, bvh_depth, bvh_leaves, scene_in_lds, lds_bytes;
    uint64_t scene_bytes;   /* bytes of nodes + triangle records + attributes on the device */
    double last_build_ms;   /* acceleration-structure build of the last trg_load_scene: host wall time, or HIP-event time when built on the GPU */
    uint32_t gpu_built, bvh_nodes4, bvh_depth4;
    uint32_t last_frame_split; /* frame lanes the last trg_render used (TRG_OPT_FRAME_SPLIT; 1 = render_kernel, 2/4 = render_fp_kernel) */
    uint32_t last_tail_bounce; /* K of the tail compaction the last trg_render used (TRG_OPT_TAIL_BOUNCE), 0 = none */
    uint32_t last_kernel;      /* enum trg_kernel the last trg_render resolved to */
    uint32_t last_regen;       /* 1: the last trg_render ran the path-regeneration megakernel (TRG_OPT_REGEN) */
    uint32_t last_tile_order;  /* the workgroup -> tile order of the last trg_render: 0 = image columns, 1/2/4/8 = XCD regions (TRG_OPT_TILE_ORDER) */
    uint32_t bvh_quads;        /* host build: pairs of triangles the builder found to be parallelograms (one test each in the shipped build) */
    uint32_t bvh_boxes;        /* ... and groups of six such quads that bound a parallelepiped and are ONE leaf of the tree the shipped build walks: twelve
                                  triangle tests in the strict build (LDS node array; the strict HBM tree keeps the subtree), one slab test in the shipped one */
} trg_stats;

enum trg_option {
    TRG_OPT_STRICT = 1,       /* 1: run the -ffp-contract=off kernels that match the oracle bit for bit (debug/parity); 0 (default): fast kernels */
    TRG_OPT_COUNTERS = 2,     /* 1: count node fetches / triangle tests (slower) */
    TRG_OPT_FORCE_GLOBAL = 3, /* 1: keep the scene in HBM even if it would fit in LDS */
    TRG_OPT_TIMING = 4,       /* 1 (default): bracket trg_render with HIP events (forces a stream sync) */
    TRG_OPT_GPU_BUILD = 6,    /* the next trg_load_scene builds the BVH on the GPU (4-wide, HBM traversal only): 1 = binned SAH, one tree level per
                                 round (the host builder's split rule), 2 = Morton order + Karras LBVH hierarchy (fastest build), 3 = Morton
                                 order + PLOC merges by surface area; 0 (default): host SAH build */
    TRG_OPT_KERNEL = 5,       /* which schedule trg_render launches: enum trg_kernel below (default TRG_KERNEL_AUTO = TRG_KERNEL_DIRECT) */
    TRG_OPT_LAUNCHES_IN_FLIGHT = 8, /* hint, default 1: how many trg_render launches of this context the caller keeps in flight on different
                                 streams (at most 16): the automatic frame split then favours throughput (the overlap hides a launch's tail).
                                 Per-launch scratch is keyed on the stream (trg_set_stream), so launches on different streams never share
                                 any, in whatever order the streams are used; at most 15 streams besides the context's own */
    TRG_OPT_TAIL_BOUNCE = 9,  /* TRG_KERNEL_DIRECT on a scene staged in LDS: K > 0 = the megakernel stops after bounce K-1, the paths that go on are
                                 compacted into per-wavefront queues in HBM (ballot / prefix, no atomics) and a second launch runs bounces
                                 K.. with every lane live again (same result bit for bit; pays when paths die over many bounces: C3).
                                 0 = never; -1 (default) = K = 2 when the launch has 4 bounces or more */
    TRG_OPT_TAIL_LEVELS = 10, /* 0 (default): the tail launches re-compact every second bounce after K; 1: one compaction at K only */
    TRG_OPT_REGEN = 11,       /* TRG_KERNEL_DIRECT on a scene in HBM (frame-serial launches): 1 = path regeneration -- every
                                 wavefront works through the (pixel, frame) jobs of its 8x8 tile as a pool, each lane at its own pace
                                 (a lane that has finished a path takes the next job instead of waiting for the slowest lane of its
                                 wavefront); the frames go through a radiance buffer that is folded in frame order, so the result is
                                 bit-identical; 0 = one pixel per lane in lock step; -1 (default) = regeneration for scenes of 32,768
                                 triangles or more (below, a shading event costs as much as its rays and lock step is faster) */
    TRG_OPT_TILE_ORDER = 12,  /* which tile of the launch a workgroup renders (any order gives the same image).  0: image columns from the centre
                                 outwards; 1, 2, 4, 8: XCD-aware -- the workgroups an XCD receives (blockIdx % 8: workgroups are dealt round-robin
                                 over the 8 XCDs, each with a private 4 MiB L2) own one of n column strips x 8/n row bands of the screen and walk
                                 it away from the image centre, so that an L2 serves the rays of one screen region (C4: 37 % less memory-side
                                 traffic, L2 hit rate 81 -> 86 %, but 14 % MORE time: the regions differ in cost and the dispatcher deals the
                                 XCDs their workgroups strictly in turn); 17, 18, 20, 24: stripes of 1, 2, 4, 8 tile rows dealt round robin over the
                                 XCDs and 32: 8 x 8-tile super-blocks with one 2 x 4-tile sub-block per XCD (round 4: balanced by construction, 4-12 %
                                 less traffic, no faster); 65, 66, 68, 72 = 64 + n: the n x (8 / n) regions dealt through ONE JOB QUEUE PER XCD --
                                 about as many persistent workgroups as the chip holds, each popping from the queue of the XCD it runs on
                                 (HW_REG_XCC_ID) and stealing from the others when it is empty (path-regeneration kernel; elsewhere = n): the
                                 traffic cut of the regions (C4: 50.9 -> 33.8 GB per launch, L2 hit rate 69 -> 77 %) without their imbalance, at
                                 +1.6 % time on C4 and -1.4 % on the 10.6 M-triangle scene; -1 (default): image columns */
    TRG_OPT_STACK_LDS_LEVELS = 13, /* scenes in HBM: levels of a thread's traversal stack kept in LDS (2..12, default 12; level 0 is the
                                 sentinel); deeper levels live in a per-launch scratch in memory.  Never changes the image: a knob for
                                 testing the scratch path and for trading LDS against it */
    TRG_OPT_TAIL_SORT = 14,   /* tail compaction: before a tail launch reads them, the queued paths of a tile are sorted by the octant of their direction
                                 (1), or by octant + the cell of their origin in a 2^3 (2) / 4^3 (3) grid over the scene box, so that the 64 lanes of a
                                 tail wavefront start alike; 0 (default) = queue order.  Same image bit for bit (a path's arithmetic is its own) */
    TRG_OPT_TAIL_REFILL = 15, /* tail compaction (round 5): 1 = the tail launches run ONE bounce each and their wavefronts REFILL -- a lane whose rays of
                                 this bounce are done takes the next queue entry instead of waiting for the slowest lane of its group of 64, lanes
                                 whose nearest-hit ray is done share a shading event -- ; 0 = lock-step groups of 64 entries, two bounces per launch.
                                 Same image bit for bit.  EXPERIMENT (experiments/lib/libtoyraygun_hip_exp.so only; the product library refuses 1):
                                 lanes per VALU instruction 24 -> 27 of 64, 24 % more time on C3 (profiles/r05/c3_tail_refill_experiment.md) */
    TRG_OPT_FRAME_SPLIT = 7   /* TRG_KERNEL_DIRECT only: frame lanes per workgroup. 1: a pixel's frames run one after the other in one
                                 lane; 2 or 4: that many wavefronts share a pixel's frames (same result bit for bit; fills the chip
                                 when the pixel grid alone does not -- small windows, row bands of a multi-GPU job);
                                 0 (default): chosen per launch from the grid size and spp.  Scenes rendered by the path-regeneration
                                 kernel (TRG_OPT_REGEN) take the lanes at workgroup level: that many workgroups share a tile's frames */
};
enum trg_kernel {
    TRG_KERNEL_DIRECT = 0,    /* one path per lane, rays traced by the lane that owns the pixel */
    TRG_KERNEL_POOL = 1,      /* workgroup path pool: ballot/prefix-compacted ray queues drained by all lanes */
    TRG_KERNEL_WAVEFRONT = 2, /* rays and path state of a batch of pixel-samples in HBM: a persistent tracer whose lanes pull the next ray of
                                 a ballot/prefix-compacted queue as soon as they are free + one shading kernel per bounce (same arithmetic
                                 and results; the schedule for scenes that live in HBM, where ray lengths diverge and paths die) */
    TRG_KERNEL_AUTO = -1      /* default: TRG_KERNEL_DIRECT for every scene (the megakernel; scenes in HBM take its path-regeneration form from 32,768
                                 triangles on, TRG_OPT_REGEN).  The wavefront schedule measured 2.5x slower on the million-triangle scene and stays an option */
};

/* --- lifetime: replaces MetalRenderer::init / resize (src/engine/Metal/MetalRenderer.mm:282-338,557-574):
 *     allocates the float4 accumulation target and the per-pixel Halton-offset texture. */
TRG_API int trg_create(trg_ctx **out, int device, uint32_t width, uint32_t height);
TRG_API void trg_destroy(trg_ctx *ctx);
TRG_API const char *trg_last_error(trg_ctx *ctx); /* ctx may be NULL: error of the last failed trg_create */

/* --- scene upload + acceleration-structure build: replaces MetalRenderer::loadScene
 *     (MetalRenderer.mm:204-280), i.e. the five memcpy at :245-249 and the MPS
 *     MPSTriangleAccelerationStructure rebuild at :272-279.  Buffers are the five public vectors of
 *     toyraygun::Scene (src/engine/Scene.h:25-29) with bx::Vec3 = 3 packed floats: positions3 has
 *     n_verts entries and is addressed through indices (3*n_tris entries); normals3 / colors3 have
 *     3*n_tris entries addressed as [triangle*3 + corner] (Raytracing.metal:104-108); material_ids has
 *     n_tris entries and doubles as the per-primitive ray mask (MetalRenderer.mm:276).  In the
 *     reference n_verts == 3*n_tris and indices is the identity.  All inputs are copied; nothing is
 *     retained. */
TRG_API int trg_load_scene(trg_ctx *ctx, const float *positions3, const float *normals3, const float *colors3,
                   const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris);

/* --- albedo textures (upstream to-do "OBJ and Texture support", README.md:18-22; Texture::loadFile, src/engine/Texture.cpp:39-48).
 *     Call after trg_load_scene; the scene then multiplies the interpolated vertex colour of a MATERIAL_DEFAULT hit by a texel.
 *     uv2: 3*n_tris (u, v) pairs addressed like the normals ([triangle*3 + corner], interpolated with the same weights);
 *     texture_ids: n_tris entries, 0 = untextured, k = images[k-1]; images: n_textures pointers to tightly packed RGBA8 rows, row 0
 *     first.  Lookup (project definition): x = min(w-1, (int)(frac(u) * w)), y = min(h-1, (int)(frac(v) * h)), texel RGB / 255,
 *     nearest, repeat.  n_tris must equal the loaded scene's; n_textures = 0 removes the textures.  Everything is copied. */
TRG_API int trg_load_textures(trg_ctx *ctx, const float *uv2, const uint32_t *texture_ids, uint32_t n_tris, const uint8_t *const *images_rgba8,
                              const uint32_t *widths, const uint32_t *heights, uint32_t n_textures);

/* --- per-frame uniforms: replaces MetalRenderer updateUniforms (MetalRenderer.mm:340-371).
 *     frameIndex in the struct is ignored by trg_render (it iterates its own range). */
TRG_API int trg_set_uniforms(trg_ctx *ctx, const trg_uniforms *u);

/* --- per-pixel Halton index offsets: replaces the R32Uint random texture filled by
 *     Texture::generateRandomTexture (src/engine/Texture.cpp:16-29; MetalRenderer.mm:315-335). */
TRG_API int trg_set_pixel_offsets(trg_ctx *ctx, const uint32_t *offsets /* width*height, host */);
TRG_API int trg_set_pixel_offsets_seed(trg_ctx *ctx, uint32_t seed); /* offset(x,y) = pcg_hash32(seed ^ (y*w+x)) on the device */

/* --- the hot path: replaces the body of MetalRenderer render: (MetalRenderer.mm:400-515): raygen, then
 *     `bounces` x [nearest-hit, primaryHit, any-hit, shadowHit], then accumulate, for frames
 *     frameIndexBegin .. frameIndexBegin+spp-1 over image rows [row0, row0+rows), in ONE launch.
 *     Result: running average in the float4 accumulation buffer (row 0 = scene bottom).  If
 *     frameIndexBegin > 0 the buffer must hold the average of frames [0, frameIndexBegin).
 *     Asynchronous when TRG_OPT_TIMING is 0.  Launches of one context into DIFFERENT accumulation buffers (trg_bind_accum) on
 *     different streams (trg_set_stream) may be in flight together (every stream gets its own traversal-stack / wavefront
 *     scratch; announce how many with TRG_OPT_LAUNCHES_IN_FLIGHT so that the schedule is chosen for throughput). */
TRG_API int trg_render(trg_ctx *ctx, uint32_t frameIndexBegin, uint32_t spp, uint32_t bounces, uint32_t row0, uint32_t rows);

/* --- INTERLEAVED row bands for multi-GPU jobs (round 4; SURVEY 8e's micro-bands): contiguous bands of the Cornell box differ in cost (the
 *     million-triangle scene at 8 GPUs: slowest band 1.2 x the mean), bands dealt round robin do not.  Rank r of n renders the 8-row
 *     MICRO-BANDS r, r + n, r + 2n ... of the image (8 rows = the sub-tile of one wavefront) and stores them COMPACTLY in rows
 *     [store_row0, store_row0 + rows) of the bound accumulation buffer: local row l is image row ((l / 8) * n + r) * 8 + l % 8.
 *     trg_microband_rows gives `rows` of a rank and the common `stride_rows` = 8 * ceil(ceil(h / 8) / n): with store_row0 = r * stride_rows
 *     the ranks' bands tile a buffer of n * stride_rows rows that an in-place all-gather completes, and trg_unpack_bands turns that
 *     compact frame into the image (a streaming kernel on the context's current stream; `image` has width*height float4, `compact`
 *     n * stride_rows rows).  Same arithmetic per pixel as trg_render: the unpacked frame is bit-identical.  Direct megakernel only. */
TRG_API void trg_microband_rows(uint32_t height, uint32_t n, uint32_t rank, uint32_t *rows, uint32_t *stride_rows); /* host-only arithmetic */
TRG_API int trg_render_bands(trg_ctx *ctx, uint32_t frameIndexBegin, uint32_t spp, uint32_t bounces, uint32_t n_ranks, uint32_t rank, uint32_t store_row0);
TRG_API int trg_unpack_bands(trg_ctx *ctx, const void *compact_device, void *image_device, uint32_t n_ranks);

/* --- read back the accumulation target (what the reference hands to its blit pass,
 *     MetalRenderer.mm:538): width*height*4 floats, host memory. */
TRG_API int trg_read_accum(trg_ctx *ctx, float *rgba);

/* ray counters are read after synchronising the context's CURRENT stream: with launches in flight on other streams
 * (TRG_OPT_LAUNCHES_IN_FLIGHT), synchronise those yourself first. */
TRG_API int trg_get_stats(trg_ctx *ctx, trg_stats *out);
TRG_API int trg_reset_stats(trg_ctx *ctx);
TRG_API int trg_set_option(trg_ctx *ctx, int option, int64_t value);

/* --- plumbing for multi-GPU / framework interop (no reference equivalent) */
TRG_API int trg_bind_accum(trg_ctx *ctx, void *device_ptr); /* use caller-owned device memory (width*height*16 B) as the accumulation buffer; NULL restores the internal one */
TRG_API int trg_accum_device_ptr(trg_ctx *ctx, void **out);
TRG_API int trg_set_stream(trg_ctx *ctx, void *hip_stream); /* NULL = the context's own stream */
TRG_API int trg_sync(trg_ctx *ctx);
/* --- asynchronous use (MetalRenderer.mm:33,377,385-387: three frames in flight behind a semaphore).  With TRG_OPT_TIMING 0
 *     trg_render only enqueues.  trg_stream_idle: 1 when everything enqueued on the context's current stream has finished, 0 when
 *     not, negative on error; never blocks.  trg_fence_record marks the current end of the stream in slot 0..7;
 *     trg_fence_wait blocks the host until the work before that mark has finished (a slot never recorded returns at once). */
TRG_API int trg_stream_idle(trg_ctx *ctx);
TRG_API int trg_fence_record(trg_ctx *ctx, int slot);
TRG_API int trg_fence_wait(trg_ctx *ctx, int slot);

/* --- multi-GPU (the reference is single-device): one process, one context per device, frames sharded by contiguous ROW
 *     BANDS, one RCCL exchange per frame over xGMI.  Device g of G renders rows [g*B, min(h, (g+1)*B)), B = ceil(h / G), into its
 *     slice of a full-frame buffer (padded to G*B rows, so that the in-place all-gather is exact for any height).  One PERSISTENT host
 *     thread per context drives the launches (started at create, handed each call through a condition variable); a host-built scene
 *     is built once and uploaded to every device.  A group of one device needs no RCCL (it is loaded with dlopen for G > 1) and no thread.
 *     NOT YET RUN ON MORE THAN ONE DEVICE: the development pool has single-GPU boxes, so n > 1 is covered by code review, the CPU
 *     rehearsals of tests/test_dist_gloo.py, and -- environment variable TRG_GROUP_FORCE_RCCL -- a group of one device that goes through
 *     RCCL all the same (dlopen, ncclCommInitAll, the in-place all-gather on the render's stream: a GPU test).
 *     Environment variable TRG_GROUP_EXCHANGE=copy (also the fallback when librccl.so cannot be loaded): the bands move by
 *     hipMemcpyPeerAsync ordered by events instead of RCCL.  Such a group needs no communicator, so its devices need not be distinct:
 *     n contexts on ONE device run the whole n > 1 machinery (bands, host threads, uploads, fences, gather) on a single-GPU box --
 *     the GPU tests do that for n = 2, 3 and 8 -- everything but the RCCL calls themselves. */
typedef struct trg_group trg_group;
enum trg_gather {
    TRG_GATHER_NONE = 0,  /* every device keeps only its own band */
    TRG_GATHER_ALL = 1,   /* in-place ncclAllGather: every device ends with the whole frame */
    TRG_GATHER_ROOT = 2   /* grouped ncclSend / ncclRecv: only device `root` ends with the whole frame */
};
TRG_API void trg_band_rows(uint32_t height, uint32_t n, uint32_t rank, uint32_t *row0, uint32_t *rows); /* host-only arithmetic */
TRG_API int trg_group_create(trg_group **out, const int *devices, int n, uint32_t width, uint32_t height);
TRG_API void trg_group_destroy(trg_group *g);
TRG_API const char *trg_group_last_error(trg_group *g); /* g may be NULL: error of the last failed trg_group_create */
TRG_API int trg_group_size(trg_group *g);
enum trg_exchange { TRG_EXCHANGE_NONE = 0, TRG_EXCHANGE_RCCL = 1, TRG_EXCHANGE_COPY = 2 };
TRG_API int trg_group_exchange(trg_group *g);          /* how the bands travel: nothing to exchange (one device), RCCL, or peer copies */
TRG_API trg_ctx *trg_group_ctx(trg_group *g, int rank); /* the context of one device (options, stats); owned by the group */
TRG_API int trg_group_load_scene(trg_group *g, const float *positions3, const float *normals3, const float *colors3, const uint32_t *indices,
                                 const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris); /* replicated on every device */
TRG_API int trg_group_load_textures(trg_group *g, const float *uv2, const uint32_t *texture_ids, uint32_t n_tris, const uint8_t *const *images_rgba8,
                                    const uint32_t *widths, const uint32_t *heights, uint32_t n_textures); /* trg_load_textures on every device */
TRG_API int trg_group_set_uniforms(trg_group *g, const trg_uniforms *u);
TRG_API int trg_group_set_pixel_offsets_seed(trg_group *g, uint32_t seed);
TRG_API int trg_group_set_option(trg_group *g, int option, int64_t value);
/* frames [frameIndexBegin, frameIndexBegin + spp) on every device's band, then the exchange `gather` (enqueued behind each device's
 * render on its stream; trg_group_sync / trg_group_read_accum wait for it) */
TRG_API int trg_group_render(trg_group *g, uint32_t frameIndexBegin, uint32_t spp, uint32_t bounces, int gather, int root);
TRG_API int trg_group_sync(trg_group *g);
/* frames in flight for a group (MetalRenderer.mm:33,377,385-387): trg_fence_record / trg_fence_wait on every device's stream */
TRG_API int trg_group_fence_record(trg_group *g, int slot);
TRG_API int trg_group_fence_wait(trg_group *g, int slot);
TRG_API int trg_group_read_accum(trg_group *g, int rank, float *rgba); /* width*height*4 floats of device `rank`'s frame buffer (of the LAST trg_group_render) */
/* --- frames in flight per device (round 4).  Consecutive trg_group_render calls that produce INDEPENDENT images (benchmark steps, the
 *     frames of an animation: frameIndexBegin = 0 each) may overlap: with depth k every device gets k frame buffers, k render streams and one
 *     exchange stream; call i renders on stream i % k into buffer i % k, its exchange follows on the exchange stream behind an event,
 *     and re-use of a buffer waits ON THE DEVICE for the exchange that read it -- the render of call i + 1 overlaps the tail of render i and
 *     exchange i (what MetalRenderer.mm:33,377,385-387 bounds to three frames).  depth 1 (default) = one buffer, the accumulation
 *     continued in place across calls (the plugin's progressive mode), the exchange on the render's stream.  The call waits for the
 *     group to drain and switches TRG_OPT_TIMING off for depth > 1 (a pipelined group only enqueues; time its launches with
 *     trg_group_time_launches).  The host side of the pipeline is bounded with the group fences. */
TRG_API int trg_group_set_pipeline(trg_group *g, int depth /* 1..8 */);
TRG_API int trg_group_pipeline_depth(trg_group *g);
/* per-launch device times without a host sync: on != 0 brackets every trg_render of every rank with HIP events on the stream it runs
 * on; trg_group_launch_ms waits for the rank's device work, returns the durations recorded since the last read (oldest first, at most
 * cap of them in ms_out; *n = how many there were) and forgets them.  Overlapping launches share the device: each lasts longer than alone. */
TRG_API int trg_group_time_launches(trg_group *g, int on);
TRG_API int trg_group_launch_ms(trg_group *g, int rank, double *ms_out, uint32_t cap, uint32_t *n);
/* --- how the rows are dealt to the devices (round 4).  TRG_BANDS_CONTIGUOUS (default): device g renders rows [g*B, (g+1)*B).
 *     TRG_BANDS_INTERLEAVED: device g renders the 8-row micro-bands g, g + G, g + 2G ... (trg_render_bands) compactly into its slice of the
 *     frame buffer, the exchange completes the compact frame and trg_unpack_bands turns it into the image on the exchange stream (one
 *     more w x h buffer per frame slot and device).  Evens out the cost of the bands (C2 at 8 GPUs: slowest / mean 1.05 -> measured in
 *     DESIGN.md; the million-triangle scene 1.2 ->).  trg_group_read_accum / trg_group_postprocess return the image either way; the
 *     contexts' own trg_read_accum sees the COMPACT buffer of an interleaved group.  Waits for the group to drain. */
enum trg_bands { TRG_BANDS_CONTIGUOUS = 0, TRG_BANDS_INTERLEAVED = 1 };
TRG_API int trg_group_set_bands(trg_group *g, int mode);
TRG_API int trg_group_bands(trg_group *g);
TRG_API int trg_group_postprocess(trg_group *g, int rank, uint8_t *rgba8, int flip_y); /* trg_postprocess of device `rank`'s last frame */
/* why the exchange is not the one asked for ("" when it is): e.g. librccl.so could not be loaded and the bands move by peer copies */
TRG_API const char *trg_group_exchange_note(trg_group *g);
/* 1 in experiments/lib/libtoyraygun_hip_exp.so -- this library plus the schedules that were built, measured and lost (TRG_KERNEL_POOL,
 * TRG_KERNEL_WAVEFRONT: experiments/README.md) --, 0 in the product library, whose trg_set_option refuses those two.  Host-only. */
TRG_API int trg_library_experiments(void);
/* what RCCL itself says about the group's communicators (round 5): *version = ncclGetVersion (e.g. 22203), ranks_out[r] = ncclCommCount of
 * device r's communicator -- the number of ranks RCCL saw --, for r < cap.  TRG_ERR_INVALID for a group without communicators (one device,
 * or the copy exchange): *version = 0 then. */
TRG_API int trg_group_rccl_info(trg_group *g, int *version, int *ranks_out, int cap);
TRG_API int trg_group_get_stats(trg_group *g, trg_stats *out);         /* ray counters summed over the devices, times of the slowest one */

/* --- stage-level entry points used by the parity tests (each isolates one SURVEY 8a row) */
/* a7 / a12: the intersector alone.  any_hit=0: out = trg_isect[n]; any_hit=1: out = float[n] distance (<0: unoccluded). */
TRG_API int trg_trace(trg_ctx *ctx, const trg_ray *rays, size_t n, int any_hit, void *out);
/* a5: halton(i[k], d[k]) */
TRG_API int trg_halton(trg_ctx *ctx, const uint32_t *i, const uint32_t *d, size_t n, float *out);
/* a5 as the SHIPPED megakernel evaluates it on an LDS-resident scene: dimensions 1..5 through the digit-group tables that are
 * staged into LDS with the scene (not bit-identical to the digit loop: within 2 ulp), every other dimension through the digit
 * code.  A table index outside [0, radix) yields NaN.  Needs a loaded LDS-sized scene; always runs the fast build. */
TRG_API int trg_halton_table(trg_ctx *ctx, const uint32_t *i, const uint32_t *d, size_t n, float *out);
/* a6: primary rays of one frame for the whole image (width*height trg_ray) */
TRG_API int trg_raygen(trg_ctx *ctx, uint32_t frameIndex, trg_ray *out);
/* a10/a11: out[k] = {dir.xyz, dist, color.xyz, 0, bounce_dir.xyz, 0} for hit point p[k], unit normal n[k],
 * random pairs r[k] = (r0,r1,r2,r3) */
TRG_API int trg_sample(trg_ctx *ctx, const float *p3, const float *n3, const float *r4, size_t n, float *out12);

/* host-only (no GPU): the acceleration structure trg_load_scene would build, flattened as DESIGN.md
 * "Data layout in HBM" describes (16 floats per node, 12 floats per triangle record).  Pass NULL
 * outputs to query the sizes. */
TRG_API int trg_debug_build_bvh(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts,
                        uint32_t n_tris, float *nodes_out, uint32_t nodes_cap, float *tris_out, uint32_t tris_cap,
                        uint32_t *n_nodes, uint32_t *n_tri_records, uint32_t *depth);

/* host-only: the 4-wide collapse of the same tree that HBM-resident scenes are traversed with (32 floats per node:
 * lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] child[4] pad[4]; unused child slot = 0x80000000). */
TRG_API int trg_debug_build_bvh4(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts,
                                 uint32_t n_tris, float *nodes4_out, uint32_t nodes4_cap, uint32_t *n_nodes4, uint32_t *depth4);

/* host-only: the BOX leaves the builder finds (scenes staged in LDS: twelve consecutive triangles that are six quads bounding a parallelepiped
 * -- Scene::addCube, src/engine/Scene.cpp:24-58 -- become one leaf: twelve triangle tests in the strict build, one slab test in the box's own
 * frame in the shipped one).  20 floats per box: node of the subtree's root and first record (in the tree trg_debug_build_bvh returns when
 * TRG_DEBUG_BVH_BOXES=1 is set), centre (3), the rows a_k of the frame (9: l_k = a_k . (P - centre), inside <=> |l_k| <= 1), and per face
 * f = 2 k + (l_k > 0) the offset of its quad's first record (6).  Pass NULL to query the count. */
TRG_API int trg_debug_boxes(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris,
                            float *boxes20_out, uint32_t boxes_cap, uint32_t *n_boxes);

/* host-only: the same 4-wide nodes in the 64-byte quantised form the HBM kernels actually load (16 dwords per node:
 * origin.xyz scale.x | qlo.x qhi.x qlo.y qhi.y | qlo.z qhi.z scale.y scale.z | child[4]; one byte per child in each
 * q dword; plane = origin + q * scale).  Node i of this array is node i of trg_debug_build_bvh4. */
TRG_API int trg_debug_build_bvh4q(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts,
                                  uint32_t n_tris, uint32_t *nodes4q_out, uint32_t nodes4_cap, uint32_t *n_nodes4);

/* host-only: the 128-byte leaf records an HBM-resident scene is traversed through, exactly as trg_load_scene lays them out (32 floats
 * per record, leaf order of trg_debug_build_bvh: rows 0-2 = (v0 | original index) (e1 | mask = material id) (e2 | 0), floats 12-20 the
 * triangle's nine normal floats, 21-29 its nine colour floats).  Pass NULL to query the count. */
TRG_API int trg_debug_leaf_records(const float *positions3, const float *normals3, const float *colors3, const uint32_t *indices,
                                   const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris, float *records32_out, uint32_t records_cap,
                                   uint32_t *n_records);

/* host-only: the PLANE form of the triangle records (round 4) -- what the shipped build's triangle test reads on an LDS-resident scene, exactly
 * as trg_load_scene lays it out: per record of trg_debug_build_bvh 12 floats (n.xyz d0 | n1.xyz d1 | n2.xyz d2: t = (d0 - n.o) / (n.dir),
 * P = o + t dir, u = n1.P + d1 = weight of vertex 1, v = n2.P + d2 = weight of vertex 2) and one u16 = (original index << 2) | (material id & 3).
 * The planes are relative to the centre of the scene's bounding box (center3_out; the ray origin is shifted by it when a traversal begins), so
 * that n.o - d0 cancels numbers of the scene's size, not of its distance from the coordinate origin.  A degenerate triangle gets planes no
 * ray passes.  Pass NULL outputs to query the count. */
TRG_API int trg_debug_plane_records(const float *positions3, const uint32_t *indices, const uint32_t *material_ids, uint32_t n_verts, uint32_t n_tris,
                                    float *planes12_out, uint16_t *meta_out, uint32_t records_cap, uint32_t *n_records, float *center3_out);

/* host-only: the tile (bx, by) workgroup slot `slot` of a launch over tiles_x x tiles_y tiles renders under tile order `order` (0, 1, 2, 4, 8:
 * TRG_OPT_TILE_ORDER), exactly as the kernels compute it.  Returns 1, 0 for a padding slot of an XCD-aware order (or a slot beyond the
 * launch), negative on bad arguments; *n_slots = workgroup slots of the launch.  The slots with return 1 cover every tile exactly once. */
TRG_API int trg_debug_tile_of_slot(uint32_t tiles_x, uint32_t tiles_y, uint32_t order, uint32_t slot, uint32_t *n_slots, uint32_t *bx, uint32_t *by);

/* which XCD runs workgroup b of a plain launch of n_blocks 256-thread workgroups: out[b] = HW_REG_XCC_ID (0..7).  The tile order above
 * relies, for speed only, on the observed round-robin placement (b and b + 8 share an XCD); this is how the tests and profiles check it. */
TRG_API int trg_debug_xcc_ids(trg_ctx *ctx, uint32_t n_blocks, uint32_t *out);

/* host-only: byte layout of the device blob of an HBM-resident scene of n_tris triangles and n_nodes4 wide nodes, computed
 * exactly as trg_load_scene does (64-bit arithmetic).  TRG_ERR_RANGE when it does not fit the 32-bit offsets the kernels
 * use (4 GiB) -- trg_load_scene refuses such a scene before it allocates or copies anything.  offsets7 (may be NULL) =
 * the LDS part (nodes, 48-byte records, normals, colours, material ids: all empty for a scene too large for LDS), the quantised
 * wide nodes, the 128-byte leaf records (geometry + attributes, one cache line per triangle). */
TRG_API int trg_debug_scene_layout(uint64_t n_tris, uint64_t n_nodes4, uint64_t *total_bytes, uint32_t *offsets7);

/* --- N1: ACES tonemap + sRGB of the accumulation buffer to RGBA8 (PostProcessing.metal:44-57;
 *     common.h:36-43,163-171).  flip_y != 0 writes the top image row first (PNG order). */
TRG_API int trg_postprocess(trg_ctx *ctx, uint8_t *rgba8, int flip_y);

#ifdef __cplusplus
}
#endif
#endif /* TRG_H */
