"""Extended randomised parity fuzz (not part of the test suite: minutes, not seconds): random soups, image sizes, spp,
bounces and option combinations; the strict build must match the oracle bit for bit, ray counts included.
python scripts/gpu_fuzz.py [cases] [seed] [fast]   (fast: the shipped build against the libm oracle within the stated tolerance,
images of at least 32x32 pixels);  FUZZ_ONLY=118,2133 replays single cases of such a run (e.g. under TRG_BVH_QUADS=0)"""
import os, sys, time; sys.path.insert(0, ".")
os.environ.setdefault("TRG_GROUP_EXCHANGE", "copy")   # device groups of several contexts on this one GPU (bands by peer copies)
import numpy as np
from toyraygun_amd import capi
from oracle import pyoracle as O
sys.path.insert(0, "tests")
from util import edge_flip_allowance

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
FAST = len(sys.argv) > 3 and sys.argv[3] == "fast"
ONLY = set(int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x.strip())
eye = np.eye(4, dtype=np.float32)
_unit = O.OracleScene(); _unit.add("cube", (1.0, 1.0, 1.0), eye)
CUBE = _unit.buffers()["positions"].reshape(-1, 3).copy()      # addCube's 36 vertices in its own order (Scene.cpp:24-58)
bad = 0
t_start = time.time()
for case in range(cases):
    kind = rng.choice(["box", "small", "large"], p=[0.2, 0.45, 0.35])
    s = O.OracleScene.cornell_box() if kind != "large" or rng.random() < 0.5 else O.OracleScene()
    n = 0 if kind == "box" else int(rng.integers(1, 150)) if kind == "small" else int(rng.integers(400, 6000))
    if n:
        ctr = rng.uniform([-0.9, 0.1, -0.9], [0.9, 1.9, 0.9], (n, 3)).astype(np.float32)
        tri = ctr[:, None, :] + rng.normal(0, rng.choice([0.02, 0.1, 0.4]), (n, 3, 3)).astype(np.float32)
        if n > 10:
            tri[n // 2: n // 2 + n // 10] = tri[: n // 10]
        mats = rng.choice([1, 1, 1, 2, 3], n)
        quads = rng.random(n) < rng.choice([0.0, 0.3, 0.8])     # (round 4: parallelograms in both index patterns of the reference's shapes -> quad leaves)
        if n > 10:
            quads[n // 2: n // 2 + n // 10] = quads[: n // 10]     # an exact duplicate of a parallelogram is a parallelogram (a lone triangle that
            # duplicates HALF of a quad would z-fight in the shipped build: the quad's plane and the triangle's round t differently -- DESIGN.md)
        for k in range(n):
            if quads[k]:
                p4 = np.concatenate([tri[k], (tri[k][0] + (tri[k][2] - tri[k][1]))[None, :]]).astype(np.float32)      # a, b, c, d = a + (c - b)
                if rng.random() < 0.5:
                    s.add_geometry(p4, [0, 1, 2, 0, 2, 3], eye, rng.uniform(0.2, 0.9, 3), int(mats[k]))
                else:
                    s.add_geometry(p4, [0, 2, 3, 0, 1, 2], eye, rng.uniform(0.2, 0.9, 3), int(mats[k]))      # (a, c, d)(a, b, c): the second is X
            else:
                s.add_geometry(tri[k], [0, 1, 2], eye, rng.uniform(0.2, 0.9, 3), int(mats[k]))
    # round 5: parallelepipeds (addCube's twelve triangles under a random rotation, anisotropic scale with random signs, sometimes a shear) of
    # every material -> BOX leaves where the scene is staged in LDS, six quads per cube everywhere else
    for k in range(int(rng.integers(1, 6)) if rng.random() < 0.5 else 0):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        sh = np.eye(3)
        if rng.random() < 0.3:
            sh[0, 1] = rng.uniform(-0.8, 0.8)
        m = np.eye(4)
        m[:3, :3] = q @ sh @ np.diag(rng.uniform(0.03, 0.3, 3) * rng.choice([-1.0, 1.0], 3))
        m[:3, 3] = rng.uniform([-0.8, 0.1, -0.8], [0.8, 1.8, 0.8])
        s.add_geometry(CUBE, np.arange(36, dtype=np.uint32), m.T.astype(np.float32), rng.uniform(0.2, 0.9, 3), int(rng.choice([1, 1, 1, 2, 3])))
    w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
    if FAST:
        w, h = max(w, 32), max(h, 32)
    spp, bnc = int(rng.integers(1, 41)) if rng.random() < 0.25 else int(rng.integers(1, 9)), int(rng.integers(0, 7))
    opts = dict(force_global=int(rng.integers(0, 2)), fsplit=int(rng.choice([0, 1, 2, 4])), gpu_build=int(rng.integers(0, 4)) if s.ntris >= 2 else 0,
                kernel=int(rng.choice([0, 0, 0, 1, 2])), in_flight=int(rng.choice([1, 4])), counters=int(rng.integers(0, 2)),
                regen=int(rng.choice([-1, 0, 1])), tail=int(rng.choice([-1, 0, 1, 2])), tile_order=int(rng.choice([-1, 0, 1, 2, 4, 8, 17, 18, 20, 24, 32, 65, 66, 68, 72])), stack_levels=int(rng.choice([12, 12, 2, 3, 6])),
                tail_sort=int(rng.choice([0, 0, 1, 2, 3])), bands=int(rng.integers(0, 2)), depth=int(rng.choice([1, 1, 2, 3])))   # (round 4: the stripe / sub-block / job-queue tile orders, the tail sort, interleaved bands and pipelined groups)
    oseed = int(rng.integers(1, 2 ** 31))
    off = O.pixel_offsets(w, h, seed=oseed)
    opts["group"] = int(rng.choice([0, 0, 0, 2, 3, 5]))   # 0: a plain context; n: a device group of n contexts on device 0 (row bands, possibly empty ones)
    if ONLY and case not in ONLY:
        # FUZZ_ONLY=case,case,...: replay single cases of a run -- the others only draw what they would have drawn
        rng.integers(0, spp + 1)
        if opts["group"] and opts["depth"] > 1:
            rng.integers(0, 4)
        if opts["group"]:
            rng.integers(0, opts["group"])
        continue
    O.set_trig_mode(O.TRIG_LIBM if FAST else O.TRIG_PORTABLE)
    ref, rst = O.render(s, w, h, spp, bnc, offsets=off)
    O.set_trig_mode(O.TRIG_LIBM)
    b = s.buffers()
    c = capi.Group([0] * opts["group"], w, h) if opts["group"] else capi.Context(w, h)
    try:
        c.set_option(capi.OPT_GPU_BUILD, opts["gpu_build"])
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        if opts["group"]:
            c.set_pixel_offsets_seed(oseed)
        else:
            c.set_pixel_offsets(off)
        c.set_option(capi.OPT_STRICT, 0 if FAST else 1)
        c.set_option(capi.OPT_FORCE_GLOBAL, opts["force_global"])
        c.set_option(capi.OPT_FRAME_SPLIT, opts["fsplit"])
        if not capi.has_experiments():
            opts["kernel"] = 0            # (the product library has the direct megakernel only; TRG_HIP_SO=experiments/lib/... fuzzes the pool / wavefront schedules too)
        c.set_option(capi.OPT_KERNEL, opts["kernel"])
        c.set_option(capi.OPT_LAUNCHES_IN_FLIGHT, opts["in_flight"])
        c.set_option(capi.OPT_COUNTERS, opts["counters"])
        c.set_option(capi.OPT_REGEN, opts["regen"])
        c.set_option(capi.OPT_TAIL_BOUNCE, opts["tail"])
        c.set_option(capi.OPT_TILE_ORDER, opts["tile_order"])
        c.set_option(capi.OPT_STACK_LDS_LEVELS, opts["stack_levels"])
        c.set_option(capi.OPT_TAIL_SORT, opts["tail_sort"])
        interleave_ok = opts["kernel"] == 0     # (interleaved bands are rendered by the direct megakernel only)
        if opts["group"]:
            c.set_bands(capi.BANDS_INTERLEAVED if opts["bands"] and interleave_ok else capi.BANDS_CONTIGUOUS)
        split = int(rng.integers(0, spp + 1))
        if opts["group"] and opts["depth"] > 1:
            # a pipelined group: independent frames in flight -- some other pictures first (their slots are re-used), the wanted one last, in one launch
            c.set_option(capi.OPT_TIMING, 0)
            c.set_pipeline(opts["depth"])
            for k in range(int(rng.integers(0, 4))):
                c.render(0, 1 + k % 2, max(bnc - 1, 0))
            split = 0
            c.render(0, spp, bnc)
        else:
            if split:
                c.render(0, split, bnc)
            if spp - split:
                c.render(split, spp - split, bnc)
        if opts["group"]:
            c.sync()
        if opts["group"] and opts["depth"] > 1:
            c.reset_stats(); c.render(0, spp, bnc); c.sync()     # (the counters of the earlier pictures are not the wanted frame's)
        img, st = (c.read_accum(int(rng.integers(0, opts["group"]))) if opts["group"] else c.read_accum()), c.stats()
        if FAST:
            d = np.linalg.norm(img[..., :3].astype(np.float64) - ref[..., :3], axis=-1)
            nr = np.linalg.norm(ref[..., :3].astype(np.float64), axis=-1)
            inl = d <= 1e-4 * np.maximum(1.0, nr)
            frac = float(np.mean(inl))
            # on these small images ONE edge-flip pixel (a ray that picks the other of two duplicate / coplanar triangles
            # under FMA rounding) already exceeds the 1e-3 RMSE meant for megapixel frames: RMSE over the inliers, and at
            # most 0.1 % outliers ("the <= 0.1 % outliers allowed are edge-flip pixels")
            # (RMSE of the error RELATIVE to max(1, |ref|), like the per-pixel bar: a soup triangle a hair away from the light gives a pixel of
            #  radiance 10^4 -- 1/distance^2 -- whose 1e-4 relative error would otherwise be the whole RMSE of a 2,000-pixel image: seed 302, case 1998)
            rel = d / np.maximum(1.0, nr)
            rmse = float(np.sqrt(np.mean((rel * rel)[inl]))) if inl.any() else 0.0
            # the outliers: 0.1 % of the pixels + the Poisson bound of the measured edge flips per ray (tests/util.py edge_flip_allowance; round 5:
            # no "small-image rule" any more)
            ok = rmse <= 1e-3 and int((~inl).sum()) <= edge_flip_allowance(w * h, rst.rays)
            if not ok:
                print("  rmse %.3g frac_ok %.5f (%d of %d pixels off, allowed %d for %d rays)" % (rmse, frac, int((d > 1e-4 * np.maximum(1.0, nr)).sum()), w * h, edge_flip_allowance(w * h, rst.rays), rst.rays))
        else:
            ok = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and st.rays == rst.rays
    finally:
        c.close()
    if not ok:
        bad += 1
        print("MISMATCH case %d: tris %d %dx%d spp %d (split %d) bounces %d %s rays %d vs %d" % (case, s.ntris, w, h, spp, split, bnc, opts, st.rays, rst.rays), flush=True)
    if case % 20 == 19:
        print("case %d, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t_start), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
