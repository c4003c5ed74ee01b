// trg_regen.inc.h -- path regeneration: the megakernel for HBM-resident scenes with every lane running its own schedule.
// Included by trg_kernels.hip after path_radiance / shade_event.
//
// render_kernel walks a wavefront's 64 pixels in lock step: every trace call ends when its slowest lane does, and on the
// million-triangle scene 16 of the 64 lanes do useful work per instruction (profiles/r02).  Here a workgroup owns the same 16x16 pixels
// and the chunk's frames of them as a POOL of path jobs (job j = frame j / 256 of tile pixel j % 256), and every lane of its four
// wavefronts works through jobs at its own pace: a lane whose rays are finished waits only until enough lanes are in the same
// situation (a shading event to run, or a path to close and a new job to take), then those lanes run that block together while the
// others keep traversing.
// Neither the spread of ray lengths inside a trace call nor the spread of cost between the pixels of a tile leaves lanes idle:
// what is left is the end of the pool.  Path state never leaves the lane (registers plus a few LDS words), jobs are handed out
// with one LDS counter per workgroup and a ballot prefix (one LDS atomic per block of lanes) -- no queue in memory, no barrier.
//
// A finished path's radiance is APPENDED to the workgroup's log in HBM -- (radiance, frame-in-chunk << 8 | pixel-in-tile), 16 bytes,
// staged 32 records at a time per wavefront in LDS and written 512 bytes at a time -- and regen_accumulate_kernel sorts a tile's log by (frame, pixel) in LDS and
// folds the chunk into the running average in frame order with Accumulate.metal's arithmetic, so the image is bit-identical to
// render_kernel's (and to the oracle's in the strict build).  (Writing each radiance to its [frame][pixel] slot instead cost 142
// bytes of memory-side write traffic per 16-byte store -- 4.7 GB per C4 launch: the eight pixels of a 128-byte line finish tens of
// microseconds apart and L2 evicts the partial line in between.)
//
// Lane states: TRAVERSING (the shadow ray, then the nearest-hit ray) or WAITING for one of these blocks:
//   class 0      close the path that just ended (fold the last shadow ray, store the radiance) and take the next job of the pool
//   class 1      a shading event (the Halton dimensions of bounce b have compile-time bases: they are evaluated bounce by bounce for
//                the lanes at that bounce, a scalar branch each; the rest of the event runs once for all lanes of the block)
// A block runs when at least TRG_REGEN_MIN lanes wait for it (TRG_REGEN_MIN0 for class 0), when more than TRG_REGEN_MAX_WAIT lanes
// wait for anything, or when nobody is traversing; the class is taken from a waiting lane picked round robin, and the waiting lanes
// are looked at every TRG_REGEN_PERIOD-th iteration only.  Measured on C4 (scripts/exp_ab.py, ms alone / per step in the pipeline;
// the lock-step kernel: 25.9 / 22.6), in the order built: a lane keeping its pixel, one class per bounce, thresholds 16 / 16 / 40
// every iteration 33.6 / 30.1 (the spread between the pixels of a tile stays); jobs from a per-wavefront pool, 8 / 8 / 16 every 2nd
// 23.4 / 20.9; 8 / 2 / 16 every 4th 22.5 / 20.5; one pool per workgroup 20.7 / 18.7; ONE shading class (per-bounce Halton inside the
// block) 20.0 / 18.3; with that 12 / 4 / 24: 19.3 / 17.8, 16 / 4 / 32: 19.2 / 17.7 (default), 20 / 4 / 32: 19.7 / 17.9, 24 / 8 / 40:
// 20.1 / 18.0, every 2nd iteration +0.5.  Letting lanes at a leaf wait until 8-16 of them share the triangle half of the step:
// +1.5-2.5 ms.
#pragma once

#ifndef TRG_REGEN_MIN
#define TRG_REGEN_MIN 16
#endif
#ifndef TRG_REGEN_MAX_WAIT
#define TRG_REGEN_MAX_WAIT 32
#endif
#ifndef TRG_REGEN_MIN0
#define TRG_REGEN_MIN0 4   // class 0 -- closing a path and taking a job -- costs a fifth of a shading event: a lower threshold
#endif
#ifndef TRG_REGEN_PERIOD
#define TRG_REGEN_PERIOD 4   // the waiting lanes are looked at every PERIOD-th iteration (a power of two)
#endif

constexpr uint32_t kRegenStage = 32u;
typedef __attribute__((address_space(3))) v4f regen_rec_t;
// the wavefront's staged records -> the workgroup's log: one LDS atomic reserves the range, lanes 0..cnt-1 write consecutive records
TRG_DEV void regen_flush(const regen_rec_t *stage, uint32_t cnt, v4f *rlog, lds_int_t *pool_done) {
    if (cnt == 0u) return;
    const uint32_t lane = lane_id_opaque();   // (opaque: otherwise the address of stage[lane] is hoisted out of the kernel's main loop into a VGPR of its own)
    int base = 0;
    if (lane == 0u) base = atomicAdd((int *)pool_done, (int)cnt);
    base = __builtin_amdgcn_readfirstlane(base);
    if (lane < cnt) rlog[(uint32_t)base + lane] = stage[lane];
}

TRG_DEV void regen_count(lds_int_t *wred, int k, bool pred) {
    const uint32_t n = wave_count(pred);
    if (n != 0u && lane_id() == 0u) wred[k] = wred[k] + (int)n;
}

static_assert(TRG_PARK_PATH && TRG_PARK_OFFSET, "render_regen_kernel uses all ten words per thread of the LDS render_kernel parks its path and its Halton offset in "
                                               "(plan_lds_as: 40 bytes per thread): nine of path state, and waves 2 and 3 stage their log records in word [9]");
template <bool COUNT, bool PERSIST = false>
__global__ __launch_bounds__(trg::kBlock, TRG_EXP_WAVES_HBM) void render_regen_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SceneView sc = scene_view<false>(p.sc, smem);
    sc.tex = p.tex;
    LdsStackT<trg::kBlock, true> stk;
    stk.set(smem, p.stack_off, p.stack.overflow, p.stack.klds);

    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // p.fsplit workgroups share a tile ("frame lanes", as in render_fp_kernel): lane fl takes the frames fl, fl + F, ... of the chunk, so
    // that a launch of few tiles (a row band, a small window) still fills the chip
    // (XCD-aware order: the F workgroups of a tile are F consecutive slots of ONE XCD, so that they share its L2)
    const uint32_t F = p.fsplit;   // wave-uniform
    // A JOB = one (tile, frame lane): job number = the workgroup slot of the plain launch.  Plain launch: this workgroup's own number,
    // once.  Persistent launch (p.xq): jobs popped from the queue of the XCD this workgroup runs on, until all eight queues are empty.
    lds_int_t *job_word = (lds_int_t *)(reinterpret_cast<int *>(smem + p.red_off) + 7);   // (word 7 of wavefront 0's counter row: free while the kernel runs)
    lds_int_t *wred = (lds_int_t *)(reinterpret_cast<int *>(smem + p.red_off) + wave * 8u);
    if (lane_id() < 8u) wred[lane_id()] = 0;   // (the ray counters of the whole workgroup lifetime; word 7 of wavefront 0 is the job hand-off, written behind a barrier)
    Counters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.wnodes = 0; cnt.wtris = 0;
  for (uint32_t round = 0; PERSIST || round == 0u; ++round) {   // (PERSIST = false: exactly one trip, the straight-line kernel)
    uint32_t job = blockIdx.x;
    if (PERSIST) {
        __syncthreads();   // everybody is done with the previous job (its LDS state, its log)
        if (threadIdx.x == 0) {
            uint32_t xcc;   // (read here, not kept: a workgroup stays on its XCD)
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
            int got = -1;
            for (uint32_t a = 0; a < trg::kXcds && got < 0; ++a) {   // own queue first, then the others in turn: at most eight atomics, then the workgroup ends
                const uint32_t q = (xcc + a) % trg::kXcds;
                const uint32_t k = atomicAdd(&p.xq[q], 1u);
                if (k < p.xq_jobs) got = (int)(q + trg::kXcds * k);
            }
            *job_word = got;
        }
        __syncthreads();
        const int got = *job_word;
        if (got < 0) break;
        job = (uint32_t)__builtin_amdgcn_readfirstlane(got);
    }
    const uint32_t kx = job / trg::kXcds;
    const uint32_t tile_id = p.xcd_cols ? (kx / F) * trg::kXcds + job % trg::kXcds : job / F;
    const uint32_t fl = p.xcd_cols ? kx % F : job % F;
    uint32_t bx, by;
    if (!block_tile(p, tile_id, bx, by)) { if (PERSIST) continue; else break; }   // (a padding slot of the XCD-aware order)
    const uint32_t x0 = bx * trg::kTileW, y0 = p.row0 + by * trg::kTileH;   // the workgroup's 16x16 tile
    const uint32_t frames_wg = p.spp > fl ? (p.spp - fl + F - 1u) / F : 0u, frames_max = (p.spp + F - 1u) / F;
    v4f *rlog = reinterpret_cast<v4f *>(p.tail_radbuf) + (size_t)job * trg::kBlock * frames_max;   // this job's log: 256 x frames records
    // per thread in LDS ([word][thread]): the Halton offset of the current job's pixel (1), throughput (3..5), radiance (6..8)
    lds_float_t *park = (lds_float_t *)(reinterpret_cast<float *>(smem + p.acc_off) + threadIdx.x);
    // the pool counter of the workgroup: the spare word [2] of thread 0
    lds_int_t *pool_next = (lds_int_t *)(reinterpret_cast<int *>(smem + p.acc_off) + 2 * trg::kBlock);
    lds_int_t *pool_done = pool_next + 1;   // records in the workgroup's log so far (word [2] of thread 1)
    if (threadIdx.x < 2) pool_next[threadIdx.x] = 0;
    // staging area of this wavefront: 32 records in the spare words [0] (waves 0, 1) and [9] (waves 2, 3) of the parked state
    regen_rec_t *stage = (regen_rec_t *)(reinterpret_cast<float *>(smem + p.acc_off) + (wave >> 1) * 9u * trg::kBlock) + (wave & 1u) * kRegenStage;
    uint32_t stage_cnt = 0u;   // wave-uniform
    __syncthreads();
#define TRG_RG_FRAME(j) (((j) / (uint32_t)trg::kBlock) * F + fl)   /* frame of the chunk */
#define TRG_RG_HIDX ((uint32_t)__float_as_int(park[trg::kBlock]) + p.frame_begin + TRG_RG_FRAME(TRG_RG_JOB))
    typedef const __attribute__((address_space(4))) trg_uniforms cu_t;
    cu_t *up = (cu_t *)__builtin_amdgcn_kernarg_segment_ptr();   // RenderParams::u is the first member
#define TRG_RG_U (*(const trg_uniforms *)up)

    // the ray counters (primary, bounce, shadow, shaded) are kept in the wavefront's row of the LDS reduction scratch, not in registers:
    // the blocks below run under conditions the compiler does not see as wave-uniform, where a loop-carried sum becomes a VGPR
    // (wred / cnt: set up once, before the job loop)
    const V3 light_color = mk(p.u.light_color[0], p.u.light_color[1], p.u.light_color[2]);
    // the number of the last bounce, as an SGPR the optimiser cannot see through (it rewrites `b + 1 == bounces` into `b == bounces - 1` and
    // kept that difference in a VGPR of its own across the main loop); with no bounces at all it matches no shading event, and there is none
    uint32_t last_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p.bounces - 1u));
    asm volatile("" : "+s"(last_b));
    const uint32_t n_jobs = (uint32_t)trg::kBlock * frames_wg;   // jobs of the workgroup's pool: its (j / 256)-th frame of tile pixel j % 256

    uint32_t jobb = 0u;           // the path this lane is working on (bits 0..23: its job) and the shading events it has had so far (bits 24..31)
#define TRG_RG_JOB (jobb & 0xFFFFFFu)
#define TRG_RG_B (jobb >> 24)
    bool running = frames_wg > 0u;
    bool fresh = true;            // no path yet: the first class-0 block only takes a job
    bool job_valid = false;       // the job's pixel lies inside the image / the band
    bool active = false, primary_ray = true;
    bool has_shadow = false, occluded = false, pending_next = false;
    bool in_shadow = false, in_next = false;   // traversing the shadow ray / the nearest-hit ray (neither: waiting); lane masks, not a VGPR
    V3 dnext = mk(0.0f, 0.0f, 1.0f), scol = mk(0.0f, 0.0f, 0.0f);
    Trav tv;
    trav_begin(sc, tv, mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 1.0f), 0.0f, 0u, stk.first());
    tv.node = kNodeDone;
    uint32_t rot = 0, it = 0;
#ifdef TRG_REGEN_GUARD
    uint32_t guard = 0;
#endif

    for (;;) {
#ifdef TRG_REGEN_GUARD
        if (++guard > (1u << 24)) break;   // bring-up only
#endif
        const bool waiting = running && !in_shadow && !in_next;
        const uint64_t wmask = __ballot(waiting);
        const uint64_t tmask = __ballot(in_shadow || in_next);
        if ((wmask | tmask) == 0ull) break;
#ifdef TRG_REGEN_DIAG   // measurement build: lanes waiting / out of work / iterations, reported through the node / triangle / iteration counters
        if (lane_id() == 0u) { wred[4] += __popcll(wmask); wred[5] += 64 - __popcll(wmask | tmask); wred[6] += 1; }
#endif
        it = (uint32_t)__builtin_amdgcn_readfirstlane((int)(it + 1u));   // (readfirstlane: keeps the loop's wave-uniform counters in SGPRs)
        if (wmask != 0ull && (tmask == 0ull || (it & (uint32_t)(TRG_REGEN_PERIOD - 1)) == 0u)) {
            const uint32_t cls = (fresh || !active || TRG_RG_B >= p.bounces) ? 0u : 1u;
            // the class of a waiting lane, round robin over the lanes so that no class starves
            rot = (uint32_t)__builtin_amdgcn_readfirstlane((int)((rot + 7u) & 63u));
            const uint64_t rolled = (wmask >> rot) | (rot ? (wmask << (64u - rot)) : 0ull);
            const int pick = (int)((uint32_t)(__ffsll((long long)rolled) - 1) + rot) & 63;
            const uint32_t csel = (uint32_t)__builtin_amdgcn_readlane((int)cls, pick);
            const bool mine = waiting && cls == csel;
            const uint32_t n_mine = (uint32_t)__popcll(__ballot(mine));
            if (n_mine >= (csel == 0u ? (uint32_t)TRG_REGEN_MIN0 : (uint32_t)TRG_REGEN_MIN) || (uint32_t)__popcll(wmask) > (uint32_t)TRG_REGEN_MAX_WAIT || tmask == 0ull) {
                asm volatile("" : "+s"(up));   // the uniforms this block needs are loaded after this point
                if (csel == 0u) {
                    {
                        // the paths that have ended: Raytracing.metal:240-241 for their last shadow ray; the frame's texel goes to the log,
                        // through the wavefront's 32-record staging area in LDS so that the log is written 512 bytes at a time
                        const bool done = mine && !fresh && job_valid;
                        const uint64_t dm = __ballot(done);
                        const uint32_t n_done = (uint32_t)__popcll(dm);
                        if (n_done != 0u) {
                            if (stage_cnt + n_done > kRegenStage) { regen_flush(stage, stage_cnt, rlog, pool_done); stage_cnt = 0u; }
                            v4f r4; r4.x = 0.0f; r4.y = 0.0f; r4.z = 0.0f; r4.w = 0.0f;
                            if (done) {
                                V3 rad = mk(park[6 * trg::kBlock], park[7 * trg::kBlock], park[8 * trg::kBlock]);
                                if (has_shadow && !occluded) rad = rad + scol;
                                r4.x = rad.x; r4.y = rad.y; r4.z = rad.z;
                                r4.w = __int_as_float((int)((TRG_RG_FRAME(TRG_RG_JOB) << 8) | (TRG_RG_JOB % (uint32_t)trg::kBlock)));
                            }
                            const uint32_t drank = mbcnt64(dm);
                            if (n_done > kRegenStage) {   // more than the stage holds at once (a whole wavefront finishing together): straight to the log
                                int dbase = 0;
                                if (done && drank == 0u) dbase = atomicAdd((int *)pool_done, (int)n_done);
                                dbase = __builtin_amdgcn_readlane(dbase, __ffsll((long long)dm) - 1);
                                if (done) rlog[(uint32_t)dbase + drank] = r4;
                            } else {
                                if (done) stage[stage_cnt + drank] = r4;
                                stage_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(stage_cnt + n_done));
                            }
                        }
                    }
                    {
                        // the next jobs of the pool, in order, to the lanes of this block
                        const uint64_t mm = __ballot(mine);
                        const uint32_t rank = mbcnt64(mm);
                        int base = 0;
                        if (mine && rank == 0u) base = atomicAdd((int *)pool_next, (int)__popcll(mm));   // one LDS atomic per block
                        base = __builtin_amdgcn_readlane(base, __ffsll((long long)mm) - 1);
                        if (mine) jobb = (uint32_t)base + rank;
                    }
                    if (mine) {
                        fresh = false; has_shadow = false; active = false; job_valid = false;
                        if (TRG_RG_JOB < n_jobs) {
                            // consecutive jobs = the pixels of one 8x8 sub-tile, sub-tile after sub-tile, frame after frame
                            const uint32_t pl = TRG_RG_JOB % (uint32_t)trg::kBlock, sub = pl >> 6;
                            const uint32_t x = x0 + (sub % (trg::kTileW / 8)) * 8u + (pl & 7u), y = y0 + (sub / (trg::kTileW / 8)) * 8u + ((pl >> 3) & 7u);
                            const uint32_t yi = image_row(p, y);   // (interleaved bands: the image row of this row of the accumulation buffer)
                            job_valid = (x < p.u.width) && (y < p.row0 + p.rows) && (yi < p.u.height);
                            if (job_valid) {
                                park[trg::kBlock] = __int_as_float((int)p.offsets[yi * p.u.width + x]);
                                V3 o, d;
                                raygen<false>(TRG_RG_U, x, yi, TRG_RG_HIDX, o, d, nullptr);
                                park[3 * trg::kBlock] = 1.0f; park[4 * trg::kBlock] = 1.0f; park[5 * trg::kBlock] = 1.0f;   // ray.color
                                park[6 * trg::kBlock] = 0.0f; park[7 * trg::kBlock] = 0.0f; park[8 * trg::kBlock] = 0.0f;   // the frame's texel
                                primary_ray = true;   // (taking the job has reset the event count)
                                active = p.bounces > 0u;   // with no bounce to trace the path is over as it starts
                                if (active) { trav_begin(sc, tv, o, d, INFINITY, 3u, stk.first()); in_next = true; pending_next = false; }
                            }
                        } else {
                            running = false;
                        }
                    }
                    regen_count(wred, 0, mine && running && job_valid);
                } else {
                    // the Halton dimensions of a shading event depend on its bounce number (compile-time bases): they are evaluated bounce
                    // by bounce for the lanes at that bounce; everything else of the event runs once for all lanes of the block
                    float r4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                    for (uint32_t bb = 0; bb < p.bounces; ++bb) {   // wave-uniform
                        const bool at = mine && TRG_RG_B == bb;
                        if (__ballot(at) == 0ull) continue;
                        if (at) {
                            uint32_t hi = TRG_RG_HIDX;
                            asm volatile("" : "+v"(hi));
                            if (bb == last_b) halton2<false>(hi, bb, r4, nullptr); else halton4<false>(hi, bb, r4, nullptr);
                        }
                    }
                    ShadeOut so; so.want_shadow = false; so.want_next = false; so.shaded = false;
                    if (mine) {
                        const bool last = TRG_RG_B == last_b;
                        V3 thr = mk(park[3 * trg::kBlock], park[4 * trg::kBlock], park[5 * trg::kBlock]);
                        V3 rad = mk(park[6 * trg::kBlock], park[7 * trg::kBlock], park[8 * trg::kBlock]);
                        if (has_shadow && !occluded) rad = rad + scol;
                        Hit h = trav_hit(tv);
                        box_hit_resolve(sc, tv, h);      // (a hit that is still a box: its triangle and weights, named once, here)
                        V3 o = tv.o, d = tv.d;
                        uint32_t rmask = primary_ray ? 3u : 1u;
                        so = shade_event<false, true>(TRG_RG_U, sc, h, tv.found, TRG_RG_B, last, TRG_RG_HIDX, o, d, thr, rad, rmask, active, light_color, r4);
                        primary_ray = rmask == 3u;
                        jobb += 1u << 24;
                        park[3 * trg::kBlock] = thr.x; park[4 * trg::kBlock] = thr.y; park[5 * trg::kBlock] = thr.z;
                        park[6 * trg::kBlock] = rad.x; park[7 * trg::kBlock] = rad.y; park[8 * trg::kBlock] = rad.z;
                        has_shadow = so.want_shadow; scol = so.scol;
                        if (so.want_shadow) {
                            trav_begin(sc, tv, o, so.sdir, so.smax, 1u, stk.first());
                            in_shadow = true; pending_next = so.want_next; dnext = d;
                        } else if (so.want_next) {
                            trav_begin(sc, tv, o, d, INFINITY, rmask, stk.first());
                            in_next = true; pending_next = false;
                        }
                        // neither: the path is over (last bounce, light, miss) and the lane waits for class 0
                    }
                    regen_count(wred, 3, so.shaded);
                    regen_count(wred, 2, so.want_shadow);
                    regen_count(wred, 1, so.want_next);
                }
            }
        }
        if (in_shadow || in_next) {
            tv.rmask = (in_next && primary_ray) ? 3u : 1u;   // re-formed every step from the lane masks: not a register across the loop
            trav_step_hbm<COUNT, trg::kBlock>(sc, tv, in_shadow, stk, cnt);
            if (tv.node == kNodeDone) {
                if (in_shadow) {
                    occluded = tv.found;
                    in_shadow = false;
                    if (pending_next) { trav_begin(sc, tv, tv.o, dnext, INFINITY, 1u, stk.first()); in_next = true; pending_next = false; }
                } else {
                    in_next = false;
                }
            }
        }
    }
    regen_flush(stage, stage_cnt, rlog, pool_done);
  }   // next job (persistent launch)
    // every wavefront has read the last hand-off word (-1) before the reduction below re-uses it (word 7 of wavefront 0's row is also where the
    // COUNT build puts that wavefront's triangle iterations): without this barrier a late wavefront could take the sum for one more job
    __syncthreads();
#ifdef TRG_REGEN_DIAG
    if (PERSIST && threadIdx.x == 0) *job_word = 0;
    __syncthreads();
#endif
#undef TRG_RG_HIDX
#undef TRG_RG_FRAME
#undef TRG_RG_JOB
#undef TRG_RG_B
#undef TRG_RG_U

    const uint32_t lane = lane_id_opaque();
    uint32_t vals[4] = { cnt.nodes, cnt.tris, cnt.wnodes, cnt.wtris };
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + p.red_off);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#ifdef TRG_REGEN_DIAG
        break;
#endif
        if (!COUNT) break;
        const uint32_t s = wave_sum(vals[k]);
        if (lane == 0) red[wave * 8 + 4 + k] = s;
    }
    __syncthreads();
#ifdef TRG_REGEN_DIAG
    if (wave == 0 && lane < 8u) {
#else
    if (wave == 0 && lane < (COUNT ? 8u : 4u)) {
#endif
        const uint32_t k = lane;
        unsigned long long s = 0;
        for (int wv = 0; wv < trg::kWaves; ++wv) s += red[wv * 8 + k];
        if (s) atomicAdd(&p.counters[(blockIdx.x % trg::kCounterSlots) * trg::kCounterWords + k], s);
    }
}

// ---- sort a tile's log by (frame, pixel) in LDS and fold the chunk into the running average in frame order (Accumulate.metal:19-39) ----
__global__ __launch_bounds__(trg::kBlock) void regen_accumulate_kernel(const trg::RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4f *stage = reinterpret_cast<v4f *>(smem);   // [frame in chunk][pixel in tile]
    uint32_t bx, by;
    if (!block_tile(p, blockIdx.x, bx, by)) return;
    const uint32_t pl = threadIdx.x, sub = pl >> 6;
    const uint32_t x = bx * trg::kTileW + (sub % (trg::kTileW / 8)) * 8u + (pl & 7u);
    const uint32_t y = p.row0 + by * trg::kTileH + (sub / (trg::kTileW / 8)) * 8u + ((pl >> 3) & 7u);
    const bool valid = (x < p.u.width) && (y < p.row0 + p.rows) && (image_row(p, y) < p.u.height);
    const uint32_t n_valid = (uint32_t)__syncthreads_count(valid ? 1 : 0);   // every valid pixel logged every frame of the chunk
    const uint32_t F = p.fsplit, frames_max = (p.spp + F - 1u) / F;
    for (uint32_t fl = 0; fl < F; ++fl) {   // the logs of the tile's frame lanes
        const uint32_t n_rec = n_valid * (p.spp > fl ? (p.spp - fl + F - 1u) / F : 0u);
        // the workgroup that rendered frame lane fl of this tile (render_regen_kernel's slot arithmetic, inverted)
        const uint32_t wg = p.xcd_cols ? ((blockIdx.x / trg::kXcds) * F + fl) * trg::kXcds + blockIdx.x % trg::kXcds : blockIdx.x * F + fl;
        const v4f *rlog = reinterpret_cast<const v4f *>(p.tail_radbuf) + (size_t)wg * trg::kBlock * frames_max;
        for (uint32_t i = threadIdx.x; i < n_rec; i += trg::kBlock) {
            const v4f r = rlog[i];
            const uint32_t code = (uint32_t)__float_as_int(r.w);
            stage[(code >> 8) * trg::kBlock + (code & 255u)] = r;
        }
    }
    __syncthreads();
    if (!valid) return;
    v4f *accum = reinterpret_cast<v4f *>(p.accum);
    const uint32_t pix = y * p.u.width + x;
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    if (p.frame_begin > 0) { const v4f a = accum[pix]; acc = mk(a.x, a.y, a.z); }
    for (uint32_t fl = 0; fl < p.spp; ++fl) {
        const v4f r4 = stage[fl * trg::kBlock + pl];
        const V3 rad = mk(r4.x, r4.y, r4.z);
        const uint32_t f = p.frame_begin + fl;
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
