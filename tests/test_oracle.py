"""CPU tests of the oracle itself: golden vectors + independent mathematical checks.
(The reference holds no tests or vectors for this path -- SURVEY section 4 -- so the independent checks are
what ties the oracle to ground truth; the goldens guard against regressions.)"""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest


def test_scene_matches_golden(O, golden, cornell):
    b = cornell.buffers()
    for k, v in b.items():
        g = golden["scene_" + k]
        assert v.shape == g.shape
        if v.dtype == np.float32:
            np.testing.assert_allclose(v, g, rtol=0, atol=1e-6)
        else:
            assert np.array_equal(v, g)


def test_scene_is_the_cornell_box(cornell):
    """36 tris / 108 verts; object order, extents, inward normals, colours, material ids."""
    b = cornell.buffers()
    assert b["positions"].shape == (108, 3) and b["material_ids"].shape == (36,)
    assert np.array_equal(b["indices"], np.arange(108, dtype=np.uint32))
    pos = b["positions"].reshape(36, 3, 3)
    nrm = b["normals"].reshape(36, 3, 3)
    col = b["colors"].reshape(36, 3, 3)
    assert np.array_equal(b["material_ids"], np.array([1] * 34 + [2] * 2, np.uint32))
    planes = {  # name: (tri range, axis, value, inward normal, colour)
        "ceiling": (24, 26, 1, 2.0, (0, -1, 0), (0.725, 0.71, 0.68)),
        "floor": (26, 28, 1, 0.0, (0, 1, 0), (0.725, 0.71, 0.68)),
        "left": (28, 30, 0, -1.0, (1, 0, 0), (0.63, 0.065, 0.05)),
        "right": (30, 32, 0, 1.0, (-1, 0, 0), (0.14, 0.491, 0.05)),
        "back": (32, 34, 2, -1.0, (0, 0, 1), (0.725, 0.71, 0.68)),
        "light": (34, 36, 1, 1.99, (0, -1, 0), (1, 1, 1)),
    }
    for name, (a, e, axis, val, n, c) in planes.items():
        np.testing.assert_allclose(pos[a:e, :, axis], val, atol=2e-6, err_msg=name)
        np.testing.assert_allclose(nrm[a:e], np.broadcast_to(np.array(n, np.float32), (e - a, 3, 3)), atol=2e-6, err_msg=name)
        np.testing.assert_allclose(col[a:e], np.broadcast_to(np.array(c, np.float32), (e - a, 3, 3)), atol=0, err_msg=name)
    light = pos[34:36].reshape(-1, 3)
    np.testing.assert_allclose([light[:, 0].min(), light[:, 0].max(), light[:, 2].min(), light[:, 2].max()],
                               [-0.25, 0.25, -0.25, 0.25], atol=2e-6)
    # boxes sit on the floor; tall box back-left, short box front-right (as in img/screenshot_metal.png)
    short, tall = pos[0:12].reshape(-1, 3), pos[12:24].reshape(-1, 3)
    assert abs(short[:, 1].min()) < 1e-6 and abs(short[:, 1].max() - 0.6) < 1e-6
    assert abs(tall[:, 1].min()) < 1e-6 and abs(tall[:, 1].max() - 1.2) < 1e-6
    assert short[:, 0].mean() > 0 and short[:, 2].mean() > 0 and tall[:, 0].mean() < 0 and tall[:, 2].mean() < 0
    # all normals unit length, and face normals point away from the box centres
    np.testing.assert_allclose(np.linalg.norm(nrm.reshape(-1, 3), axis=1), 1.0, atol=1e-6)
    for rng_ in ((0, 12), (12, 24)):
        ctr = pos[rng_[0]:rng_[1]].reshape(-1, 3).mean(0)
        for t in range(*rng_):
            assert np.dot(pos[t].mean(0) - ctr, nrm[t, 0]) > 0


def test_halton_matches_golden_and_exact_radical_inverse(O, golden):
    hi = golden["halton_i"]
    table = np.array([[O.halton(i, d) for d in range(64)] for i in hi], np.float32)
    assert np.array_equal(table.view(np.uint32), golden["halton"].view(np.uint32))
    # independent check: exact rational radical inverse, fp32 recurrence error is tiny
    for i in (1, 2, 3, 12345, 2 ** 24 + 1, 2 ** 32 - 1):
        for d in (0, 1, 2, 7, 15, 16, 33, 63):
            b = int(O.lib().orc_halton_prime(d))
            x, f, n = Fraction(0), Fraction(1, b), i
            while n:
                x += f * (n % b)
                n //= b
                f /= b
            assert abs(O.halton(i, d) - float(x)) < 4e-7
    assert [int(O.lib().orc_halton_prime(d)) for d in range(16)] == [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53]
    assert O.halton(0, 0) == 0.0 and O.halton(1, 0) == 0.5 and O.halton(1, 1) == np.float32(1.0) / np.float32(3.0) * 1.0


def test_pixel_offsets_are_the_seeded_pcg_hash(O, golden):
    off = O.pixel_offsets(16, 16)
    assert np.array_equal(off, golden["offsets_16x16"])

    def pcg(v):
        s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
        return ((w >> 22) ^ w) & 0xFFFFFFFF
    for y, x in ((0, 0), (3, 5), (15, 15)):
        assert int(off[y * 16 + x]) == pcg(0x5EED0001 ^ (y * 16 + x))


def test_uniforms_golden_and_camera_model(O, golden):
    for (w, h, key) in ((256, 256, "uniforms_256"), (1920, 1080, "uniforms_1920x1080")):
        raw = np.frombuffer(O.uniforms_bytes(O.make_uniforms(w, h)), np.uint8)
        a, b = raw.view(np.float32), golden[key].view(np.float32)
        assert np.array_equal(raw[:16], golden[key][:16])
        np.testing.assert_allclose(a[4:], b[4:], rtol=0, atol=2e-6)
    # the inverse really inverts view*proj; the centre pixel looks down -z from the eye
    u = O.make_uniforms(1024, 768)
    inv = np.array(u.inv_view_proj, np.float64).reshape(4, 4).T  # back to bx row-major
    L = O.lib()
    fp = C.POINTER(C.c_float)
    view, proj, vp = (np.zeros(16, np.float32) for _ in range(3))
    eye, at, up = np.array(O.EYE, np.float32), np.array(O.AT, np.float32), np.array([0, 1, 0], np.float32)
    L.orc_mtx_look_at_rh(view.ctypes.data_as(fp), eye.ctypes.data_as(fp), at.ctypes.data_as(fp), up.ctypes.data_as(fp))
    L.orc_mtx_proj_rh(proj.ctypes.data_as(fp), 45.0, 1024 / 768, 1.0, 125.0)
    L.orc_mtx_mul(vp.ctypes.data_as(fp), view.ctypes.data_as(fp), proj.ctypes.data_as(fp))
    np.testing.assert_allclose(vp.reshape(4, 4).astype(np.float64) @ inv, np.eye(4), atol=2e-4)
    w4 = np.array([0, 0, 0, 1.0]) @ inv
    d = w4[:3] / w4[3] - np.array(O.EYE)
    np.testing.assert_allclose(d / np.linalg.norm(d), [0, 0, -1], atol=1e-6)
    # vertical half-angle 22.5 degrees: top-centre NDC (0,1) maps to tan(22.5 deg) above the axis
    w4 = np.array([0, 1.0, 0, 1.0]) @ inv
    d = w4[:3] / w4[3] - np.array(O.EYE)
    assert abs(d[1] / -d[2] - np.tan(np.radians(22.5))) < 1e-5
    assert tuple(u.light_pos)[:3] == (0.0, np.float32(1.98), 0.0) and tuple(u.light_right)[:3] == (0.25, 0.0, 0.0)


def test_mtx_srt_is_scale_rotate_translate(O):
    L = O.lib()
    fp = C.POINTER(C.c_float)
    m = np.zeros(16, np.float32)
    L.orc_mtx_srt(m.ctypes.data_as(fp), 2.0, 3.0, 4.0, 0.3, -0.7, 1.1, 5.0, 6.0, 7.0)
    M = m.reshape(4, 4).astype(np.float64)

    def rx(a): return np.array([[1, 0, 0], [0, np.cos(a), np.sin(a)], [0, -np.sin(a), np.cos(a)]])
    def ry(a): return np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]])
    def rz(a): return np.array([[np.cos(a), np.sin(a), 0], [-np.sin(a), np.cos(a), 0], [0, 0, 1]])
    R = M[:3, :3] / np.array([[2.0], [3.0], [4.0]])
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)      # a rotation, after removing the row scale
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-6)
    np.testing.assert_allclose(M[3], [5, 6, 7, 1], atol=0)
    # row-vector convention: some ordering of the three axis rotations reproduces it
    cands = [a @ b @ c for a in (rx(0.3),) for b in (ry(-0.7),) for c in (rz(1.1),)] + \
            [rz(1.1) @ ry(-0.7) @ rx(0.3), ry(-0.7) @ rx(0.3) @ rz(1.1), rx(0.3) @ rz(1.1) @ ry(-0.7)]
    assert min(np.abs(R - c).max() for c in cands + [c.T for c in cands]) < 1e-6


def test_sampling_matches_golden_and_is_well_formed(O, golden):
    from tests.golden import make_golden as MG
    r, n, p = golden["sample_r"], golden["sample_n"], golden["sample_p"]
    u = O.make_uniforms(256, 256)
    for mode, tag in ((O.TRIG_LIBM, "libm"), (O.TRIG_PORTABLE, "portable")):
        O.set_trig_mode(mode)
        hemi, aligned, ldir, lcol, ldist = MG.sampling_outputs(r, n, p, u)
        np.testing.assert_allclose(hemi, golden["sample_hemi_" + tag], rtol=0, atol=1e-6)
        np.testing.assert_allclose(aligned, golden["sample_aligned_" + tag], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.linalg.norm(aligned, axis=1), 1.0, atol=2e-6)
        assert ((aligned * n).sum(1) > -1e-6).all()          # in the normal's hemisphere
        np.testing.assert_allclose(hemi[:, 1], np.sqrt(r[:, 1]), atol=1e-6)  # cos(theta) = sqrt(u.y)
    O.set_trig_mode(O.TRIG_LIBM)
    np.testing.assert_allclose(ldir, golden["sample_ldir"], atol=1e-6)
    np.testing.assert_allclose(lcol, golden["sample_lcol"], atol=1e-6)
    np.testing.assert_allclose(ldist, golden["sample_ldist"], atol=1e-6)
    # light sample: direction is unit and points at a point of the light quad
    hit = p + ldir * ldist[:, None]
    assert (np.abs(hit[:, 1] - 1.98) < 1e-5).all() and (np.abs(hit[:, 0]) <= 0.25 + 1e-5).all() and (np.abs(hit[:, 2]) <= 0.25 + 1e-5).all()


def test_portable_sincos_is_close_to_libm(O):
    L = O.lib()
    s, c = C.c_float(), C.c_float()
    phis = np.concatenate([np.linspace(0, 2 * np.pi, 20001), [0.0, np.pi / 4, np.pi / 2, np.pi, 1.5 * np.pi, 6.2831855]]).astype(np.float32)
    err = 0.0
    for phi in phis:
        L.orc_sincos_portable(float(phi), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(phi))), abs(c.value - np.cos(np.float64(phi))))
    assert err < 2.5e-7


def test_raygen_golden_and_geometry(O, golden):
    for f in (0, 5):
        rays = O.raygen(16, 16, f)
        g = golden["raygen_16x16_f%d" % f].view(O.RAY_DTYPE)
        np.testing.assert_allclose(rays["direction"], g["direction"], atol=1e-6)
        assert np.array_equal(rays["origin"], g["origin"]) and (rays["mask"] == 3).all() and np.isinf(rays["maxDistance"]).all()
    rays = O.raygen(16, 16, 0).reshape(16, 16)
    np.testing.assert_allclose(np.linalg.norm(rays["direction"], axis=-1), 1.0, atol=2e-6)
    assert (rays["direction"][..., 2] < 0).all()
    assert (np.diff(rays["direction"][..., 1].mean(axis=1)) > 0).all()   # row 0 = scene bottom, y grows with row
    assert (np.diff(rays["direction"][..., 0].mean(axis=0)) > 0).all()   # x grows with column


def _random_rays(O, n, seed, inside=True):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, O.RAY_DTYPE)
    rays["origin"] = rng.uniform([-0.95, 0.05, -0.95], [0.95, 1.9, 2.5 if not inside else 0.95], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["mask"] = rng.choice([1, 3], n).astype(np.uint32)
    rays["maxDistance"] = np.where(rng.random(n) < 0.2, rng.uniform(0.1, 2.0, n), np.inf).astype(np.float32)
    rays["maxDistance"][rng.random(n) < 0.02] = -1.0
    return rays


def test_intersector_against_double_precision_geometry(O, cornell):
    rays = _random_rays(O, 4000, 11)
    got = O.intersect_nearest(cornell, rays)
    prim, t, margin = O.nearest_f64(cornell, rays)
    clear = margin > 1e-4   # rays whose outcome does not hinge on fp32 rounding
    assert clear.mean() > 0.97
    assert np.array_equal(got["primitiveIndex"][clear], prim[clear])
    hit = clear & (prim >= 0)
    np.testing.assert_allclose(got["distance"][hit], t[hit], rtol=2e-5, atol=2e-6)
    assert (got["distance"][clear & (prim < 0)] < 0).all()
    inactive = rays["maxDistance"] < 0
    assert inactive.any() and (got["distance"][inactive] < 0).all() and (got["primitiveIndex"][inactive] == -1).all()
    # barycentric weights reproduce the hit point
    b = cornell.buffers()
    P = b["positions"].reshape(-1, 3, 3)[got["primitiveIndex"][hit]]
    c = got["coordinates"][hit]
    w = np.stack([c[:, 0], c[:, 1], 1 - c[:, 0] - c[:, 1]], 1)
    np.testing.assert_allclose((w[:, :, None] * P).sum(1), rays["origin"][hit] + rays["direction"][hit] * got["distance"][hit][:, None], atol=2e-5)
    # any-hit agrees with nearest on occlusion
    anyd = O.intersect_any(cornell, rays)
    assert np.array_equal(anyd >= 0, got["distance"] >= 0)
    # the light (mask 2) is invisible to mask-1 rays and visible to mask-3 rays
    up = np.zeros(2, O.RAY_DTYPE)
    up["origin"] = [0, 1, 0]; up["direction"] = [0, 1, 0]; up["maxDistance"] = np.inf; up["mask"] = [1, 3]
    r = O.intersect_nearest(cornell, up)
    assert r["primitiveIndex"][0] in (24, 25) and abs(r["distance"][0] - 1.0) < 1e-6
    assert r["primitiveIndex"][1] in (34, 35) and abs(r["distance"][1] - 0.99) < 1e-6


def test_oracle_bvh_equals_brute_force(O):
    s = O.OracleScene.cornell_lattice(5)   # 36 + 12*125 = 1536 triangles -> the oracle's BVH path
    assert s.ntris == 1536
    rays = _random_rays(O, 3000, 5)
    a, b = O.intersect_nearest(s, rays), O.intersect_nearest(s, rays, brute=True)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert np.array_equal(O.intersect_any(s, rays) >= 0, O.intersect_any(s, rays, brute=True) >= 0)


def test_frames_match_golden(O, golden, cornell):
    img, st = O.render(cornell, 256, 256, 1, 1)
    np.testing.assert_allclose(img, golden["frame_c1_256_1spp_1b"], rtol=0, atol=2e-6)
    assert [st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits] == golden["rays_c1"].tolist()
    for mode, tag in ((O.TRIG_LIBM, "libm"), (O.TRIG_PORTABLE, "portable")):
        O.set_trig_mode(mode)
        img, st = O.render(cornell, 64, 64, 4, 3)
        d = np.abs(img - golden["frame_64_4spp_3b_" + tag])
        assert (d.max(-1) > 1e-5).mean() < 0.002       # libm versions may differ in the last ulp -> rare edge flips
        img, _ = O.render(cornell, 48, 32, 2, 8)
        assert (np.abs(img - golden["frame_48x32_2spp_8b_" + tag]).max(-1) > 1e-5).mean() < 0.005
    O.set_trig_mode(O.TRIG_LIBM)


def test_frame_semantics(O, cornell):
    """Accumulate.metal running average, alpha, emissive overwrite, bands, continuation."""
    w = h = 64
    full, st = O.render(cornell, w, h, 6, 3)
    assert (full[..., 3] == 1.0).all() and np.isfinite(full).all() and (full[..., :3] >= 0).all()
    # continuation: frames [0,6) == frames [0,2) then [2,6)
    part, _ = O.render(cornell, w, h, 2, 3)
    part, _ = O.render(cornell, w, h, 4, 3, frame_begin=2, accum=part)
    assert np.array_equal(full, part)
    # bands: rows [0,24) + [24,64) == full
    acc = np.zeros((h, w, 4), np.float32)
    O.render(cornell, w, h, 6, 3, row0=0, rows=24, accum=acc)
    O.render(cornell, w, h, 6, 3, row0=24, rows=40, accum=acc)
    assert np.array_equal(full, acc)
    # a pixel looking straight at the light is exactly the light colour
    big, _ = O.render(cornell, 256, 256, 2, 3)
    ys, xs = np.where((big[..., :3] == 1.0).all(-1))
    assert len(ys) > 100 and ys.mean() > 128      # the light is in the upper half (row 0 = bottom)
    # thread count does not change the image
    one, _ = O.render(cornell, w, h, 6, 3, nthreads=1)
    assert np.array_equal(full, one)
    # ray accounting: every primary ray that hits default geometry spawns one shadow ray per shaded hit
    assert st.primary_rays == w * h * 6 and st.shadow_rays == st.shaded_hits and st.bounce_rays <= st.shaded_hits


def test_trig_modes_agree_within_tolerance(O, cornell):
    from tests.util import image_metrics, TOL_RMSE
    O.set_trig_mode(O.TRIG_LIBM)
    a, _ = O.render(cornell, 96, 96, 8, 3)
    O.set_trig_mode(O.TRIG_PORTABLE)
    b, _ = O.render(cornell, 96, 96, 8, 3)
    O.set_trig_mode(O.TRIG_LIBM)
    rmse, frac_ok, _ = image_metrics(b, a)
    assert rmse <= TOL_RMSE and frac_ok >= 0.995


def test_postprocess(O):
    acc = np.zeros((2, 3, 4), np.float32)
    acc[0, 0, :3] = [0, 0.18, 1.0]
    acc[1, 2, :3] = [10.0, 0.001, 0.5]
    out = O.postprocess(acc, flip_y=False)

    def ref(x):
        a, b, c, d, e = 2.51, 0.03, 2.43, 0.59, 0.14
        v = min(max((x * (a * x + b)) / (x * (c * x + d) + e), 0.0), 1.0)
        v = v * 12.92 if v < 0.0031308 else 1.055 * v ** (1 / 2.4) - 0.055
        return int(min(max(v, 0), 1) * 255 + 0.5)
    assert [int(v) for v in out[0, 0, :3]] == [ref(0), ref(0.18), ref(1.0)] and out[0, 0, 3] == 255
    assert [int(v) for v in out[1, 2, :3]] == [ref(10.0), ref(0.001), ref(0.5)]
    assert np.array_equal(O.postprocess(acc, flip_y=True), out[::-1])


def test_tuned_cpu_baseline_build_agrees_with_the_checker(O, cornell, built):
    """bench.py's `tuned` CPU-baseline leg is a second build of the SAME source (oracle/Makefile liboracle_tuned.so: -O3
    -march=x86-64-v3, contraction allowed, constant-base Halton) that walks THE PRODUCT'S BVH2 (trg_debug_build_bvh, host-only) on every
    scene.  It is never the checker -- it is checked: (1) the checker build walking the product's tree gives the checker's own frame BIT
    FOR BIT with the same ray counts (the tree changes which triangles are tested, not what a test returns; ties go to the lower index
    either way), Cornell box and a 2,628-triangle lattice; (2) the tuned build stays within the shipped build's tolerance of it and
    counts the same rays within 1e-4; (3) its constant-base Halton is the reference recurrence up to the contraction of r + f * digit."""
    from toyraygun_amd import capi
    from tests.util import TOL_FRAC, TOL_RMSE, image_metrics
    for scene, (w, h, spp, bnc) in ((cornell, (160, 120, 3, 3)), (O.OracleScene.cornell_lattice(6), (96, 64, 2, 3))):
        ref, rst = O.render(scene, w, h, spp, bnc)
        b = scene.buffers()
        nodes, tris, _ = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
        O.set_bvh(scene, nodes, tris)
        try:
            got, st = O.render(scene, w, h, spp, bnc)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and st.rays == rst.rays and st.shaded_hits == rst.shaded_hits
            assert st.node_visits > 0 and st.tri_tests < rst.tri_tests      # it really walked the tree
            rays = np.zeros(4000, O.RAY_DTYPE)
            rng = np.random.default_rng(9)
            rays["origin"] = rng.uniform((-0.9, 0.1, -0.9), (0.9, 1.9, 2.5), (4000, 3)).astype(np.float32)
            d = rng.normal(size=(4000, 3)); rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
            rays["mask"] = rng.choice([1, 2, 3], 4000).astype(np.uint32); rays["maxDistance"] = np.inf
            walked, walked_any = O.intersect_nearest(scene, rays), O.intersect_any(scene, rays)
            O.set_bvh(scene)
            assert np.array_equal(walked.view(np.uint8), O.intersect_nearest(scene, rays, brute=True).view(np.uint8))
            assert np.array_equal(walked_any >= 0, O.intersect_any(scene, rays, brute=True) >= 0)
            O.set_bvh(scene, nodes, tris)
            tun, tst = O.render(scene, w, h, spp, bnc, tuned=True)
            rmse, frac_ok, worst = image_metrics(tun, ref)
            assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
            assert tst.primary_rays == rst.primary_rays and abs(tst.rays - rst.rays) <= 1e-4 * rst.rays
        finally:
            O.set_bvh(scene)
    T = O.tuned_lib()
    import ctypes as C
    T.orc_halton.argtypes = [C.c_uint32, C.c_uint32]; T.orc_halton.restype = C.c_float
    for d in range(64):
        for i in (0, 1, 2, 17, 12345, 2 ** 24 - 1, 2 ** 24 + 1, 2 ** 31 + 7, 2 ** 32 - 1):
            a, b = float(T.orc_halton(i, d)), float(O.halton(i, d))      # (contraction: r + f * digit is one fma in the tuned build)
            assert abs(a - b) <= 2.5e-7 * max(b, 1e-3), (i, d, a, b)
