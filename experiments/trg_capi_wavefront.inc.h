// experiments/trg_capi_wavefront.inc.h -- EXPERIMENT: the host side of the wavefront schedule (TRG_KERNEL_WAVEFRONT), included by
// toyraygun_amd/csrc/trg_capi.cpp when TRG_EXPERIMENTS=1 (experiments/build.py); the kernels: experiments/trg_wavefront.inc.h.
// TRG_KERNEL_WAVEFRONT: the frames of the launch in batches of at most kWfMaxPaths pixel-samples; per batch
//   raygen, bounces x [persistent trace, shade + compacting append], one more trace for the last shadow rays, accumulate.
// Nothing synchronises with the host: queue lengths stay on the device.
static int render_wavefront(trg_ctx *c, uint32_t frame_begin, uint32_t spp, uint32_t bounces, uint32_t row0, uint32_t rows, int slot) {
    LdsPlan plan;
    if (int rc = plan_lds(c, plan, false, 0, false)) return rc;   // the tracer's LDS: scene (if staged) + stacks only
    const uint64_t npix = (uint64_t)c->w * rows;
    const uint32_t fb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spp, kWfMaxPaths / std::max<uint64_t>(npix, 1)));
    const uint64_t nb_max = npix * fb;
    if (nb_max >= (1ull << 31)) return fail(c, TRG_ERR_RANGE, "trg_render: %llu pixels in one band are too many for the wavefront schedule", (unsigned long long)npix);
    // layout of the slot: 7 float4 arrays, two lists of 2 nb entries, the stage counters
    const size_t arr = (size_t)nb_max * 16u, lst = (size_t)nb_max * 2u * 4u;
    const size_t need = 7u * arr + 2u * lst + kWfMaxStages * 16u + 256u;
    if (need > c->wf_bytes[slot]) {
        if (c->wf_mem[slot]) { (void)hipDeviceSynchronize(); (void)hipFree(c->wf_mem[slot]); c->wf_mem[slot] = nullptr; c->wf_bytes[slot] = 0; }
        hipError_t e = hipMalloc((void **)&c->wf_mem[slot], need);
        if (e != hipSuccess) return fail(c, TRG_ERR_NOMEM, "wavefront buffers hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        c->wf_bytes[slot] = need;
    }
    unsigned char *m = c->wf_mem[slot];
    WfParams p{};
    p.u = c->u; p.sc = c->sc; p.offsets = c->offsets; p.accum = c->accum; p.counters = c->counters;
    p.tex = c->tex;
    p.b.ray_o = m; p.b.ray_d = m + arr; p.b.sh = m + 2 * arr; p.b.hit = m + 3 * arr; p.b.thr = m + 4 * arr; p.b.rad = m + 5 * arr; p.b.scol = m + 6 * arr;
    p.b.list[0] = reinterpret_cast<uint32_t *>(m + 7 * arr); p.b.list[1] = reinterpret_cast<uint32_t *>(m + 7 * arr + lst);
    p.b.ctr = reinterpret_cast<uint32_t *>(m + 7 * arr + 2 * lst);
    p.npix = (uint32_t)npix; p.pix0 = row0 * c->w; p.bounces = bounces;
    p.stack_off = plan.stack_off;
    // the persistent tracer: as many workgroups as the chip holds at 8 waves/SIMD; late ones find the queue empty
    const uint32_t trace_grid = (uint32_t)c->cu_count * 8u;
    if (int rc = ensure_stack_scratch(c, plan, (uint64_t)trace_grid * kBlock, p.stack, slot)) return rc;
    const uint32_t shade_grid = (uint32_t)c->cu_count * 8u;
    const bool strict = c->opt_strict;
    for (uint32_t f0 = 0; f0 < spp; f0 += fb) {
        p.frame0 = frame_begin + f0;
        p.nframes = std::min(fb, spp - f0);
        p.nb = p.npix * p.nframes;
        HIPCHK(c, hipMemsetAsync(p.b.ctr, 0, kWfMaxStages * 16u, c->stream));
        HIPCHK(c, strict ? launch_wf_raygen_strict(p, c->stream) : launch_wf_raygen_fast(p, c->stream));
        for (uint32_t b = 0; b <= bounces; ++b) {
            if (bounces == 0) break;
            p.stage = b; p.bounce = b;
            HIPCHK(c, strict ? launch_wf_trace_strict(p, plan.lds_scene, c->opt_counters, trace_grid, plan.total, c->stream)
                             : launch_wf_trace_fast(p, plan.lds_scene, c->opt_counters, trace_grid, plan.total, c->stream));
            if (b == bounces) break;   // that was the trace of the last bounce's shadow rays
            HIPCHK(c, strict ? launch_wf_shade_strict(p, plan.lds_scene, shade_grid, c->stream) : launch_wf_shade_fast(p, plan.lds_scene, shade_grid, c->stream));
        }
        HIPCHK(c, strict ? launch_wf_accumulate_strict(p, c->stream) : launch_wf_accumulate_fast(p, c->stream));
    }
    return TRG_OK;
}


