// experiments/trg_rtail.inc.h -- EXPERIMENT, not in the shipped library (TRG_EXPERIMENTS=1 only): the tail of deep paths with IN-WAVE REFILL
// (TRG_OPT_TAIL_REFILL; round-4 verdict, item 4).  Bit-exact; lanes per VALU instruction 24.0 -> at most 27.9 of 64 and 24 % more time on C3 --
// profiles/r05/c3_tail_refill_experiment.md.  Included by toyraygun_amd/csrc/trg_tail.inc.h inside namespace trgk_{fast,strict}.
#pragma once

// ---- tail with IN-WAVE REFILL (TRG_OPT_TAIL_REFILL; round 5, the round-4 verdict's item 4): ONE bounce per launch, and a wavefront does not
//      walk its queue segment 64 entries at a time in lock step: every lane runs  [take the next entry] -> nearest-hit ray -> shading event ->
//      shadow ray -> [radiance out, survivor appended]  at its own pace.  The traversal loop is left as soon as kRtailRefill lanes have run out
//      of ray (not when the slowest lane has): lanes whose shadow ray is done close their path and take the next entries of the segment (ballot /
//      mbcnt prefix over the idle lanes, a wave-uniform cursor: no atomics), lanes whose nearest-hit ray is done wait until kRtailShade of
//      them can run the shading event together -- every path of a launch is at the SAME bounce, so its Halton dimensions keep their
//      compile-time bases.  Same per-path arithmetic in the same order: bit-identical to render_tail_kernel and to the oracle.
#ifndef TRG_RTAIL_REFILL
#define TRG_RTAIL_REFILL 16   // leave the traversal loop when this many of the lanes that entered it have finished their ray
#endif
#ifndef TRG_RTAIL_SHADE
#define TRG_RTAIL_SHADE 32    // run the shading event when this many lanes wait for it (or nothing else is left to do)
#endif
#ifndef TRG_EXP_WAVES_RTAIL
#define TRG_EXP_WAVES_RTAIL 7   // a lane's traversal state has to survive the shading event of OTHER lanes: 64 VGPRs spill 59 of them, 72 do not
#endif
template <bool COUNT>
__global__ __launch_bounds__(trg::kBlock, TRG_EXP_WAVES_RTAIL) void render_rtail_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SceneView sc = scene_view<true>(p.sc, smem);
    sc.tex = p.tex;
    LdsStackT<trg::kBlock, false> stk;
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t W = blockIdx.x * (uint32_t)trg::kWaves + wave;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.tail_count[W]);
    const v4f *queue = reinterpret_cast<const v4f *>(p.tail_queue) + (size_t)W * p.tail_cap * 3u;
    v4f *queue_out = reinterpret_cast<v4f *>(p.tail_queue_out) + (size_t)W * p.tail_cap * 3u;
    v4f *radbuf = reinterpret_cast<v4f *>(p.tail_radbuf);
    const uint32_t b = p.tail_k;                          // the bounce of this launch (wave-uniform)
    const bool last = b + 1u == p.bounces;                // ... the last one: nothing is queued again
    constexpr bool TAB = !TRG_STRICT && TRG_HALTON_TABLES;
    typedef const __attribute__((address_space(4))) trg_uniforms cu_t;
    cu_t *up = (cu_t *)__builtin_amdgcn_kernarg_segment_ptr();
    PathCounters pc; pc.primary = 0; pc.bounce = 0; pc.shadow = 0; pc.shaded = 0;
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);

    uint32_t next = 0u, fill = 0u;                        // wave-uniform: entries taken from the segment / appended to the output segment
    // lane state: in_t1 / in_t2 = traversing the nearest-hit / the shadow ray, wait_s = nearest-hit ray done, shading event pending; none: idle
    bool in_t1 = false, in_t2 = false, wait_s = false, primary_ray = false;
    V3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f), x = mk(0.0f, 0.0f, 0.0f), rad = mk(0.0f, 0.0f, 0.0f);   // x: throughput until the shading event, then the pending shadow colour
    uint32_t pl = 0u, fl = 0u, hidx = 0u;
    Trav tv;
    trav_begin(sc, tv, o, d, 0.0f, 0u, stk.first(), lds_node_base<false>(sc), true);
    tv.node = kNodeDone;
    for (;;) {
        // ---- refill: idle lanes take the next entries of the segment ----
        const uint64_t idle = __ballot(!(in_t1 || in_t2 || wait_s));
        if (next < n && idle != 0ull) {
            const uint32_t my = next + mbcnt64(idle);
            const bool take = !(in_t1 || in_t2 || wait_s) && my < n;
            if (take) {
                const v4f e0 = queue[(size_t)my * 3u], e1 = queue[(size_t)my * 3u + 1u], e2 = queue[(size_t)my * 3u + 2u];
                o = mk(e0.x, e0.y, e0.z); d = mk(e1.x, e1.y, e1.z); x = mk(e2.x, e2.y, e2.z);
                pl = __float_as_uint(e0.w);
                const uint32_t fbits = __float_as_uint(e1.w);
                fl = fbits & 0x7FFFFFFFu; primary_ray = (fbits >> 31) != 0u;
                const v4f r4 = radbuf[(size_t)fl * p.tail_band_pixels + pl];
                rad = mk(r4.x, r4.y, r4.z);
                uint32_t offset;
                if (p.il_n <= 1u) {
                    offset = p.offsets[p.row0 * p.u.width + pl];
                } else {   // interleaved bands: `pl` counts pixels of the compact band; the Halton offsets are indexed by image pixel
                    const uint32_t l = pl / p.u.width;
                    offset = p.offsets[image_row(p, p.row0 + l) * p.u.width + (pl - l * p.u.width)];
                }
                hidx = offset + p.frame_begin + fl;
                trav_begin(sc, tv, o, d, INFINITY, primary_ray ? 3u : 1u, stk.first(), lds_node_base<false>(sc), true);
                in_t1 = true;
            }
            const uint32_t took = (uint32_t)__popcll(__ballot(take));
            next += took;
            if (b > 0) pc.bounce += took;
        }
        const uint64_t tmask = __ballot(in_t1 || in_t2), wmask = __ballot(wait_s);
        if ((tmask | wmask) == 0ull) break;              // the segment is used up and every lane is idle
        // ---- the shading event of this bounce, for the lanes that wait for it ----
        if (wmask != 0ull && ((uint32_t)__popcll(wmask) >= (uint32_t)TRG_RTAIL_SHADE || tmask == 0ull)) {
            asm volatile("" : "+s"(up));
            ShadeOut so; so.want_shadow = false; so.want_next = false; so.shaded = false; so.sdir = mk(0.0f, 0.0f, 1.0f); so.scol = mk(0.0f, 0.0f, 0.0f); so.smax = -1.0f;
            bool active = false;
            if (wait_s) {
                Hit h = trav_hit(tv);
                lds_box_hit_resolve(sc, tv, h);      // (a hit that is still a BOX: its triangle and weights)
                uint32_t rmask = primary_ray ? 3u : 1u;
                active = true;
                so = shade_event<TAB>(*(const trg_uniforms *)up, sc, h, tv.found, b, last, hidx, o, d, x, rad, rmask, active, light_color);
                primary_ray = rmask == 3u;
            }
            pc.shaded += wave_count(so.shaded);
            pc.shadow += wave_count(so.want_shadow);
            // the paths that go on: ballot / mbcnt prefix into this wavefront's segment of the output queue
            if (!last) {
                const bool go = wait_s && active;
                const uint64_t m = __ballot(go);
                if (m != 0ull) {
                    if (go) {
                        v4f *e = queue_out + (size_t)(fill + mbcnt64(m)) * 3u;
                        v4f e0, e1, e2;
                        e0.x = o.x; e0.y = o.y; e0.z = o.z; e0.w = __uint_as_float(pl);
                        e1.x = d.x; e1.y = d.y; e1.z = d.z; e1.w = __uint_as_float(fl | (primary_ray ? 0x80000000u : 0u));
                        e2.x = x.x; e2.y = x.y; e2.z = x.z; e2.w = 0.0f;
                        e[0] = e0; e[1] = e1; e[2] = e2;
                    }
                    fill += (uint32_t)__popcll(m);
                }
            }
            if (wait_s) {
                wait_s = false;
                if (so.want_shadow) {
                    x = so.scol;
                    trav_begin(sc, tv, o, so.sdir, so.smax, 1u, stk.first(), lds_node_base<false>(sc), true);
                    in_t2 = true;
                } else {                                  // no shadow ray (a miss, the light, a light sample behind the surface): the path's frame value is final for this bounce
                    v4f r4; r4.x = rad.x; r4.y = rad.y; r4.z = rad.z; r4.w = 0.0f;
                    radbuf[(size_t)fl * p.tail_band_pixels + pl] = r4;
                }
            }
        }
        // ---- traverse: every lane with a ray, until TRG_RTAIL_REFILL of them have finished (or all) ----
        if (in_t1 || in_t2) {
            const uint32_t entered = (uint32_t)__popcll(__ballot(1));
            // (leaving early only pays while the finished lanes have something to do: entries to take, or a shading event they help to fill)
            const uint32_t stay = (next < n || (uint32_t)__popcll(wmask) + (uint32_t)TRG_RTAIL_REFILL >= (uint32_t)TRG_RTAIL_SHADE) && entered > (uint32_t)TRG_RTAIL_REFILL ? entered - (uint32_t)TRG_RTAIL_REFILL : 0u;
            for (;;) {
                while (tv.node >= 0) trav_node_step_signed<COUNT, trg::kBlock>(sc, tv, stk, cnt);
                if (tv.node == kNodeDone) break;
                if (trav_leaf_step<COUNT, trg::kBlock>(sc, tv, in_t2, stk, cnt)) { tv.node = kNodeDone; break; }
                if (tv.node == kNodeDone) break;
                if ((uint32_t)__popcll(__ballot(1)) <= stay) break;
            }
            if (tv.node == kNodeDone) {
                if (in_t1) {
                    in_t1 = false; wait_s = true;
                } else {
                    in_t2 = false;
                    if (!tv.found) rad = rad + x;        // Raytracing.metal:240-241
                    v4f r4; r4.x = rad.x; r4.y = rad.y; r4.z = rad.z; r4.w = 0.0f;
                    radbuf[(size_t)fl * p.tail_band_pixels + pl] = r4;
                }
            }
        }
    }
    if (!last && lane_id_opaque() == 0) p.tail_count_out[W] = fill;
    flush_counters(p, smem, wave, lane_id_opaque(), pc, cnt, COUNT);
}

