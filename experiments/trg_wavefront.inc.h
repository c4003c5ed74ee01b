// experiments/trg_wavefront.inc.h -- EXPERIMENT, not in the shipped library: the WAVEFRONT schedule of the same path tracer
// (TRG_KERNEL_WAVEFRONT), included by toyraygun_amd/csrc/trg_kernels.hip inside namespace trgk_{fast,strict} when TRG_EXPERIMENTS=1
// (experiments/build.py).  Bit-exact against the oracle; 2.5x slower than the megakernel on C4, 18x on C2 -- NOTEBOOK.md, "Wavefront schedule".
//
// Why a second schedule.  The megakernel gives a path to one lane for its whole life: perfect when the scene sits in LDS and a
// ray is a few dozen instructions away from its hit, poor when the scene lives in HBM.  On the 1 M-triangle scene (C4) the
// lanes of a wavefront finish their traversals at very different times (18 node steps per ray on average, several times that
// for the slowest lane) and paths die along the way: profiles/r02 measures 16 of 64 lanes active per VALU instruction, and the
// kernel is bound by the texture-addresser path, which is paid per wavefront instruction whatever the number of live lanes.
// Here the rays of a batch of pixel-samples live in HBM (what the reference itself does: Raytracing.metal's ray / shadow-ray /
// intersection buffers, MetalRenderer.mm:285-297), and
//   * wf_trace_kernel is a PERSISTENT tracer: every lane of every resident wavefront pulls the next ray of a compacted queue as
//     soon as its own ray is finished (one atomic per wavefront per refill, slots handed out with a ballot / mbcnt prefix), so
//     the traversal loop runs with nearly all lanes busy whatever the spread of ray lengths;
//   * wf_shade_kernel runs the shading event of one bounce for the paths whose nearest-hit ray was just traced and appends the
//     continuation and shadow rays that are really needed to the next queue with the same wave-level ballot / prefix compaction:
//     dead paths simply stop appearing, so the next trace pass is as long as the rays that are left.
// Arithmetic per path is the megakernel's (same raygen, traverse steps, shade_event, accumulate order): the strict build is
// bit-identical to the oracle and to render_kernel.  HBM traffic: 112 bytes of path state, about 250 bytes moved per path and
// bounce -- a few GB per C4 frame, against the 2x FETCH_SIZE = 53 GB the traversal itself pulls through L2.
//
// Path state, structure of arrays indexed by path id = frame_in_batch * npix + pixel_in_band (all float4):
//   ray_o = (origin.xyz, maxDistance)     ray_d = (direction.xyz, bits mask)            the nearest-hit ray to trace next
//   sh    = (shadow direction.xyz, shadow maxDistance or -1)   origin = ray_o.xyz;  the tracer sets .w = -1 when occluded,
//           so .w >= 0 after the trace means "add scol" (shadowHit, Raytracing.metal:219-246)
//   hit   = (t or -1, bits primitiveIndex, u, v)      thr, rad, scol = throughput, radiance so far, pending shadow colour
// Queue entry = path id | kind << 31 (kind 1 = shadow ray, any-hit, mask 1).  Stage s has counters ctr[4 s] = head (next entry
// to hand out) and ctr[4 s + 1] = entries; stage 0 is the implicit identity list of the primary rays.

// Path state is streamed once per stage: non-temporal loads / stores keep it from displacing the scene (the BVH the tracer
// re-reads all the time) in L2 and the Infinity Cache.
#ifndef TRG_WF_LINEAR
#define TRG_WF_LINEAR 0
#endif
#ifndef TRG_WF_NT
#define TRG_WF_NT 1
#endif
TRG_DEV v4f wf_ld(const v4f *p) {
#if TRG_WF_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
TRG_DEV void wf_st(v4f *p, v4f v) {
#if TRG_WF_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
struct WfView {
    v4f *ray_o, *ray_d, *sh, *hit, *thr, *rad, *scol;
};
TRG_DEV WfView wf_view(const trg::WfParams &p) {
    WfView v;
    v.ray_o = reinterpret_cast<v4f *>(p.b.ray_o); v.ray_d = reinterpret_cast<v4f *>(p.b.ray_d); v.sh = reinterpret_cast<v4f *>(p.b.sh);
    v.hit = reinterpret_cast<v4f *>(p.b.hit); v.thr = reinterpret_cast<v4f *>(p.b.thr); v.rad = reinterpret_cast<v4f *>(p.b.rad);
    v.scol = reinterpret_cast<v4f *>(p.b.scol);
    return v;
}

// pixel of a path: the band's pixels are contiguous in the row-major image (pix0 = row0 * width)
// Path id -> pixel.  64 consecutive ids are an 8x8 pixel tile (as in the megakernel: neighbouring lanes trace neighbouring rays),
// tiles row-major over the band; the ragged right / bottom edge of a band whose size is not a multiple of 8 falls back to the
// row-major order for the whole band.  The accumulate kernel uses the same map.
TRG_DEV uint32_t wf_local_pixel(const trg::WfParams &p, uint32_t pl) {
    const uint32_t w = p.u.width, rows = p.npix / w;
    if (((w | rows) & 7u) != 0u || TRG_WF_LINEAR) return pl;
    const uint32_t tile = pl >> 6, in = pl & 63u, tx = tile % (w >> 3), ty = tile / (w >> 3);
    return (ty * 8u + (in >> 3)) * w + tx * 8u + (in & 7u);
}
TRG_DEV void wf_pixel(const trg::WfParams &p, uint32_t pid, uint32_t &pix, uint32_t &x, uint32_t &y, uint32_t &frame) {
    const uint32_t fl = pid / p.npix, pl = pid - fl * p.npix;
    pix = p.pix0 + wf_local_pixel(p, pl);
    y = pix / p.u.width; x = pix - y * p.u.width;
    frame = p.frame0 + fl;
}

// ---- raygen (Raytracing.metal:41-91) for every path of the batch ----
__global__ __launch_bounds__(256) void wf_raygen_kernel(const trg::WfParams p) {
    const WfView w = wf_view(p);
    const uint32_t pid = blockIdx.x * 256u + threadIdx.x;
    const bool valid = pid < p.nb;
    if (valid) {
        uint32_t pix, x, y, frame;
        wf_pixel(p, pid, pix, x, y, frame);
        V3 o, d;
        raygen(p.u, x, y, p.offsets[pix] + frame, o, d);
        v4f r0, r1, one, zero, nosh;
        r0.x = o.x; r0.y = o.y; r0.z = o.z; r0.w = INFINITY;
        r1.x = d.x; r1.y = d.y; r1.z = d.z; r1.w = __int_as_float(3);  // RAY_MASK_PRIMARY
        one.x = 1.0f; one.y = 1.0f; one.z = 1.0f; one.w = 0.0f;
        zero.x = 0.0f; zero.y = 0.0f; zero.z = 0.0f; zero.w = 0.0f;
        nosh.x = 0.0f; nosh.y = 0.0f; nosh.z = 1.0f; nosh.w = -1.0f;
        wf_st(&w.ray_o[pid], r0); wf_st(&w.ray_d[pid], r1); wf_st(&w.thr[pid], one); wf_st(&w.rad[pid], zero); wf_st(&w.sh[pid], nosh);
    }
    const uint32_t n = wave_count(valid);
    if (lane_id() == 0 && n) atomicAdd(&p.counters[(blockIdx.x % trg::kCounterSlots) * trg::kCounterWords + 0], (unsigned long long)n);
}

// ---- the persistent tracer ----
#ifndef TRG_WF_REFILL_MIN
#define TRG_WF_REFILL_MIN 16
#endif
#ifndef TRG_WF_SLICE_MAX
#define TRG_WF_SLICE_MAX 512u
#endif
template <bool LDS_SCENE, bool COUNT>
__global__ __launch_bounds__(trg::kBlock, 8) void wf_trace_kernel(const trg::WfParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SceneView sc = scene_view<LDS_SCENE>(p.sc, smem);
    LdsStackT<trg::kBlock, !LDS_SCENE> stk;
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);
    const WfView w = wf_view(p);
    const uint32_t *list = p.stage ? p.b.list[p.stage & 1u] : nullptr;   // stage 0: the identity list of the primary rays
    const uint32_t count = p.stage ? p.b.ctr[4u * p.stage + 1u] : p.nb;
    uint32_t *head = &p.b.ctr[4u * p.stage];
    const uint32_t lane = threadIdx.x & 63u;
    constexpr bool UNIFIED = !LDS_SCENE;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;

    bool busy = false, exhausted = false, any = false;
    uint32_t pid = 0;
    Trav tv;
    trav_begin(sc, tv, mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 1.0f), 0.0f, 0u, stk.first(), lds_node_base<UNIFIED>(sc), !UNIFIED);
    tv.node = kNodeDone;
    // This wavefront's reserved slice [cur, end) of the queue.  A slice is taken with ONE atomic on the queue head and then handed
    // out to idle lanes with a ballot / mbcnt prefix, no memory traffic: one word sustains only ~90 atomics per microsecond
    // (MI355X_MICROARCH.md, "dequeue"), and a refill per atomic made the first version of this kernel 3.5x slower than the
    // megakernel.  Slices shrink towards the end of the queue so that its tail is spread over the wavefronts.
    uint32_t cur = 0, end = 0;
    const uint32_t nwaves = gridDim.x * (trg::kBlock / 64u);
    for (;;) {
        // ---- refill: idle lanes take the next entries of the slice ----
        const uint64_t idle = __ballot(!busy);
        if (!exhausted && idle != 0ull) {
            if (cur == end) {   // wave-uniform: reserve the next slice
                uint32_t base = 0, want = 0;
                if (lane == 0) {
                    const uint32_t seen = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t rem = seen < count ? count - seen : 0u;
                    want = rem / (2u * nwaves);
                    want = want < 64u ? 64u : (want > TRG_WF_SLICE_MAX ? TRG_WF_SLICE_MAX : want);
                    base = atomicAdd(head, want);
                }
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                want = (uint32_t)__builtin_amdgcn_readfirstlane((int)want);
                cur = base < count ? base : count;
                end = base + want < count ? base + want : count;
                if (cur == end) exhausted = true;
            }
            const uint32_t my = cur + mbcnt64(idle);
            if (!busy && my < end) {
                const uint32_t e = list ? list[my] : my;
                pid = e & 0x7FFFFFFFu;
                any = (e >> 31) != 0u;
                const v4f r0 = wf_ld(&w.ray_o[pid]);
                const v4f r1 = any ? wf_ld(&w.sh[pid]) : wf_ld(&w.ray_d[pid]);
                trav_begin(sc, tv, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), any ? r1.w : r0.w, any ? 1u : (uint32_t)__float_as_int(r1.w),
                           stk.first(), lds_node_base<UNIFIED>(sc), !UNIFIED);
                busy = true;
            }
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            cur = cur + n_idle < end ? cur + n_idle : end;
        }
        if (__ballot(busy) == 0ull) break;
        if (busy) {
            for (;;) {
                if (UNIFIED) {
                    trav_step_hbm<COUNT, trg::kBlock>(sc, tv, any, stk, cnt);
                } else {
                    while (tv.node >= 0) trav_node_step_signed<COUNT, trg::kBlock>(sc, tv, stk, cnt);
                    if (tv.node == kNodeDone) break;
                    if (trav_leaf_step<COUNT, trg::kBlock>(sc, tv, any, stk, cnt)) tv.node = kNodeDone;
                }
                if (tv.node == kNodeDone) break;
                // enough lanes of this wavefront have run dry: let them refill (this lane's state stays in registers)
                if (!exhausted && __popcll(__ballot(1)) <= 64 - TRG_WF_REFILL_MIN) break;
            }
            if (tv.node == kNodeDone) {
                if (any) {
                    if (tv.found) reinterpret_cast<float *>(&w.sh[pid])[3] = -1.0f;   // occluded: nothing to add
                } else {
                    v4f h;
                    Hit bh = trav_hit(tv);      // (a hit that is still a BOX gets its triangle and weights here: trg_device.h box_hit_resolve / lds_box_hit_resolve)
                    box_hit_resolve(sc, tv, bh); lds_box_hit_resolve(sc, tv, bh);
                    h.x = bh.t; h.y = __int_as_float(bh.prim); h.z = bh.u; h.w = bh.v;
                    wf_st(&w.hit[pid], h);
                }
                busy = false;
            }
        }
    }
    if (COUNT) {
        uint32_t vals[4] = { cnt.nodes, cnt.tris, cnt.wnodes, cnt.wtris };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t s = wave_sum(vals[k]);
            if (lane == 0 && s) atomicAdd(&p.counters[((blockIdx.x * 4u + (threadIdx.x >> 6)) % trg::kCounterSlots) * trg::kCounterWords + 4 + k], (unsigned long long)s);
        }
    }
}

// Workgroup-compacted append of up to two entries per thread (continuation ray, shadow ray) to a global queue: ballot / mbcnt
// prefix inside each wavefront, the four wavefront totals combined through LDS, ONE atomic per workgroup (a per-wavefront
// atomic on the one tail word was the second bottleneck of the first version).  Called by all 256 threads, uniformly.
TRG_DEV void wf_append2(bool pa, uint32_t va, bool pb, uint32_t vb, uint32_t *list, uint32_t *counter, uint32_t *lds8) {
    const uint64_t ma = __ballot(pa), mb = __ballot(pb);
    const uint32_t na = (uint32_t)__popcll(ma), nb_ = (uint32_t)__popcll(mb);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0) lds8[wave] = na + nb_;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t0 = lds8[0], t1 = lds8[1], t2 = lds8[2], t3 = lds8[3];
        const uint32_t tot = t0 + t1 + t2 + t3;
        const uint32_t base = tot ? atomicAdd(counter, tot) : 0u;
        lds8[4] = base; lds8[5] = base + t0; lds8[6] = base + t0 + t1; lds8[7] = base + t0 + t1 + t2;
    }
    __syncthreads();
    const uint32_t wbase = lds8[4 + wave];
    if (pa) list[wbase + mbcnt64(ma)] = va;
    if (pb) list[wbase + na + mbcnt64(mb)] = vb;
    __syncthreads();   // lds8 is reused by the next call
}

// ---- the shading event of bounce p.bounce (primaryHit, Raytracing.metal:115-215; the shadowHit of the previous bounce first) for
//      every nearest-hit entry of stage p.stage; appends the rays of stage p.stage + 1 ----
// FAT: the tracer walked the scene from HBM, so the hit records name leaf records (trg_device.h kRecV4) and the attributes come from there
template <bool FAT>
__global__ __launch_bounds__(256) void wf_shade_kernel(const trg::WfParams p) {
    const WfView w = wf_view(p);
    const trg::SceneDesc &sd = p.sc;
    SceneView sc;
    sc.nodes = nullptr; sc.htab = nullptr; sc.meta = nullptr; sc.center = mk(sd.center[0], sd.center[1], sd.center[2]);
    sc.tex = p.tex;
    sc.tris = FAT ? reinterpret_cast<const v4f *>(sd.blob + (kRecPlanes ? sd.off_fat_planes : sd.off_fat)) : nullptr;
    sc.normals = FAT ? nullptr : reinterpret_cast<const float *>(sd.blob + sd.off_normals);
    sc.colors = FAT ? nullptr : reinterpret_cast<const float *>(sd.blob + sd.off_colors);
    sc.mats = FAT ? nullptr : reinterpret_cast<const uint32_t *>(sd.blob + sd.off_mats);
    const uint32_t *list = p.stage ? p.b.list[p.stage & 1u] : nullptr;
    const uint32_t count = p.stage ? p.b.ctr[4u * p.stage + 1u] : p.nb;
    uint32_t *next = p.b.list[(p.stage + 1u) & 1u];
    uint32_t *n_next = &p.b.ctr[4u * (p.stage + 1u) + 1u];
    const uint32_t b = p.bounce;
    const bool last = (b + 1u == p.bounces);
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);
    uint32_t n_bounce = 0, n_shadow = 0, n_shaded = 0;
    __shared__ uint32_t lds8[8];
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t base = blockIdx.x * 256u; base < count; base += stride) {   // workgroup-uniform trip count (wf_append2 has barriers)
        const uint32_t i = base + threadIdx.x;
        const uint32_t e = i < count ? (list ? list[i] : i) : 0x80000000u;
        const uint32_t pid = e & 0x7FFFFFFFu;
        ShadeOut so; so.want_shadow = false; so.want_next = false; so.shaded = false;
        if ((e >> 31) == 0u) {
            uint32_t pix, x, y, frame;
            wf_pixel(p, pid, pix, x, y, frame);
            const v4f r0 = wf_ld(&w.ray_o[pid]), r1 = wf_ld(&w.ray_d[pid]), hv = wf_ld(&w.hit[pid]), tv4 = wf_ld(&w.thr[pid]), rv4 = wf_ld(&w.rad[pid]);
            v4f shv = wf_ld(&w.sh[pid]);
            V3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z), thr = mk(tv4.x, tv4.y, tv4.z), rad = mk(rv4.x, rv4.y, rv4.z);
            if (shv.w >= 0.0f) {   // shadowHit of the previous bounce: its ray was traced in this stage and found nothing
                const v4f sc4 = wf_ld(&w.scol[pid]);
                rad = rad + mk(sc4.x, sc4.y, sc4.z);
            }
            uint32_t rmask = (uint32_t)__float_as_int(r1.w);
            bool active = true;
            Hit h; h.t = hv.x; h.prim = __float_as_int(hv.y); h.u = hv.z; h.v = hv.w;
            const bool found = hv.x >= 0.0f;
            so = shade_event<false, FAT>(p.u, sc, h, found, b, last, p.offsets[pix] + frame, o, d, thr, rad, rmask, active, light_color);
            v4f t4, ra4;
            t4.x = thr.x; t4.y = thr.y; t4.z = thr.z; t4.w = 0.0f;
            ra4.x = rad.x; ra4.y = rad.y; ra4.z = rad.z; ra4.w = 0.0f;
            wf_st(&w.thr[pid], t4); wf_st(&w.rad[pid], ra4);
            shv.x = so.sdir.x; shv.y = so.sdir.y; shv.z = so.sdir.z; shv.w = so.want_shadow ? so.smax : -1.0f;
            wf_st(&w.sh[pid], shv);
            if (so.want_shadow) { v4f c4; c4.x = so.scol.x; c4.y = so.scol.y; c4.z = so.scol.z; c4.w = 0.0f; wf_st(&w.scol[pid], c4); }
            if (so.shaded) {   // the origin of both new rays (a path that hit the invalid material keeps its ray as it is)
                v4f q0, q1;
                q0.x = o.x; q0.y = o.y; q0.z = o.z; q0.w = INFINITY;
                q1.x = d.x; q1.y = d.y; q1.z = d.z; q1.w = __int_as_float((int)rmask);
                wf_st(&w.ray_o[pid], q0);
                if (so.want_next) wf_st(&w.ray_d[pid], q1);
            }
        }
        wf_append2(so.want_next, pid, so.want_shadow, pid | 0x80000000u, next, n_next, lds8);
        n_bounce += wave_count(so.want_next); n_shadow += wave_count(so.want_shadow); n_shaded += wave_count(so.shaded);
    }
    if (lane_id() == 0) {
        unsigned long long *c = &p.counters[((blockIdx.x * 4u + (threadIdx.x >> 6)) % trg::kCounterSlots) * trg::kCounterWords];
        if (n_bounce) atomicAdd(&c[1], (unsigned long long)n_bounce);
        if (n_shadow) atomicAdd(&c[2], (unsigned long long)n_shadow);
        if (n_shaded) atomicAdd(&c[3], (unsigned long long)n_shaded);
    }
}

// ---- last shadowHit + accumulate (Accumulate.metal:19-39), the frames of the batch in order ----
__global__ __launch_bounds__(256) void wf_accumulate_kernel(const trg::WfParams p) {
    const WfView w = wf_view(p);
    const uint32_t pl = blockIdx.x * 256u + threadIdx.x;
    if (pl >= p.npix) return;
    v4f *accum = reinterpret_cast<v4f *>(p.accum);
    const uint32_t pix = p.pix0 + wf_local_pixel(p, pl);
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    if (p.frame0 > 0) { const v4f a = accum[pix]; acc = mk(a.x, a.y, a.z); }
    for (uint32_t fl = 0; fl < p.nframes; ++fl) {
        const uint32_t pid = fl * p.npix + pl;
        const v4f rv4 = wf_ld(&w.rad[pid]), shv = wf_ld(&w.sh[pid]);
        V3 rad = mk(rv4.x, rv4.y, rv4.z);
        if (shv.w >= 0.0f) { const v4f c4 = wf_ld(&w.scol[pid]); rad = rad + mk(c4.x, c4.y, c4.z); }   // the path's last shadow ray found nothing
        const uint32_t f = p.frame0 + fl;
        if (f == 0) {
            acc = rad;
        } else {
            const V3 prev = acc * (float)f;
            const V3 c = rad + prev;
            const float f1 = (float)(f + 1u);
            acc = mk(c.x / f1, c.y / f1, c.z / f1);
        }
    }
    v4f outv; outv.x = acc.x; outv.y = acc.y; outv.z = acc.z; outv.w = 1.0f;
    accum[pix] = outv;
}
