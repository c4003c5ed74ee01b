"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Bars (north_star / SURVEY 8d):
  * integer / index work and the STRICT build (-ffp-contract=off): bit-exact against the oracle
    (oracle in ORC_TRIG_PORTABLE mode, the device's sin/cos definition);
  * the shipped FAST build (FMA contraction, v_rcp/v_rsq): per-pixel L2 on the linear float4 buffer,
    RMSE <= 1e-3 and >= 99.9 % of pixels within 1e-4 * max(1, |ref|), against the oracle in its
    faithful libm mode.
Nothing here reads /root/reference.
"""
import os
import numpy as np
import pytest

from tests.util import TOL_FRAC, TOL_FRAC_C4, TOL_RMSE, edge_flip_allowance, image_metrics, make_ctx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(built):
    from toyraygun_amd import capi as c
    c.load()
    return c


@pytest.fixture(scope="module")
def ctx256(capi, O, cornell):
    c = make_ctx(O, cornell, 256, 256, offsets=O.pixel_offsets(256, 256))
    yield c
    c.close()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ------------------------------------------------------------------ a5 Halton
def test_halton_bit_exact(capi, O, ctx256, golden):
    rng = np.random.default_rng(3)
    i = np.concatenate([golden["halton_i"], rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32)])
    ii, dd = np.repeat(i, 64), np.tile(np.arange(64, dtype=np.uint32), i.shape[0])
    ref = np.array([O.halton(a, b) for a, b in zip(ii.tolist(), dd.tolist())], np.float32)
    for strict in (0, 1):
        ctx256.set_option(capi.OPT_STRICT, strict)
        got = ctx256.halton(ii, dd)
        assert np.array_equal(_bits(got), _bits(ref)), "halton differs (strict=%d)" % strict
    n = golden["halton_i"].shape[0]
    assert np.array_equal(_bits(got[: n * 64].reshape(n, 64)), _bits(golden["halton"]))
    ctx256.set_option(capi.OPT_STRICT, 0)
    with pytest.raises(capi.TrgError):
        ctx256.halton([1], [64])
    assert ctx256.halton([], []).shape == (0,)


# ------------------------------------------------------------------ a6 raygen
def test_raygen(capi, O, ctx256):
    off = O.pixel_offsets(256, 256)
    for f in (0, 3, 2 ** 32 - 1):   # the last one wraps offset + frameIndex mod 2^32
        ref = np.zeros(256 * 256, O.RAY_DTYPE)
        u = O.make_uniforms(256, 256, f)
        L = O.lib()
        import ctypes as C
        dst = np.zeros((256, 256, 4), np.float32)
        for y in range(0, 256, 5):
            for x in range(0, 256, 3):
                L.orc_raygen(C.byref(u), off.ctypes.data, ref.ctypes.data, dst.ctypes.data, x, y)
        sel = np.zeros((256, 256), bool)
        sel[0:256:5, 0:256:3] = True
        sel = sel.reshape(-1)
        ctx256.set_option(capi.OPT_STRICT, 1)
        got = ctx256.raygen(f)
        assert np.array_equal(got[sel].view(np.uint8), ref[sel].view(np.uint8)), "strict raygen differs at frame %d" % f
        ctx256.set_option(capi.OPT_STRICT, 0)
        got = ctx256.raygen(f)
        np.testing.assert_allclose(got["direction"][sel], ref["direction"][sel], rtol=0, atol=3e-7)
        assert np.array_equal(got["origin"][sel], ref["origin"][sel]) and (got["mask"] == 3).all() and np.isinf(got["maxDistance"]).all()


# ------------------------------------------------------------------ a10 / a11 sampling
def test_sampling(capi, O, ctx256, golden):
    r, n, p = golden["sample_r"], golden["sample_n"], golden["sample_p"]
    r4 = np.concatenate([r, r[::-1]], axis=1).astype(np.float32)
    ctx256.set_option(capi.OPT_STRICT, 1)
    got = ctx256.sample(p, n, r4)
    assert np.array_equal(_bits(got[:, 0:3]), _bits(golden["sample_ldir"]))
    assert np.array_equal(_bits(got[:, 3]), _bits(golden["sample_ldist"]))
    assert np.array_equal(_bits(got[:, 4:7]), _bits(golden["sample_lcol"]))
    # bounce direction for (r2, r3) = reversed grid, against the oracle's portable-trig definition
    import ctypes as C
    L = O.lib()
    fp = C.POINTER(C.c_float)
    O.set_trig_mode(O.TRIG_PORTABLE)
    ref = np.zeros((r.shape[0], 3), np.float32)
    for k in range(r.shape[0]):
        rr, nn, h = r4[k, 2:4].copy(), n[k].copy(), np.zeros(3, np.float32)
        L.orc_sample_cosine_hemisphere(rr.ctypes.data_as(fp), h.ctypes.data_as(fp))
        L.orc_align_hemisphere(h.ctypes.data_as(fp), nn.ctypes.data_as(fp), ref[k].ctypes.data_as(fp))
    O.set_trig_mode(O.TRIG_LIBM)
    assert np.array_equal(_bits(got[:, 8:11]), _bits(ref))
    ctx256.set_option(capi.OPT_STRICT, 0)
    fast = ctx256.sample(p, n, r4)
    np.testing.assert_allclose(fast, got, rtol=2e-6, atol=2e-6)


# ------------------------------------------------------------------ a7 / a12 intersector
def _rays(O, n, seed, lo=(-0.95, 0.05, -0.95), hi=(0.95, 1.9, 3.0)):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, O.RAY_DTYPE)
    rays["origin"] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["mask"] = rng.choice([1, 2, 3], n).astype(np.uint32)
    rays["maxDistance"] = np.where(rng.random(n) < 0.25, rng.uniform(0.05, 3.0, n), np.inf).astype(np.float32)
    rays["maxDistance"][rng.random(n) < 0.02] = -1.0
    return rays


def _adversarial_rays(O, scene):
    """Rays aimed at shared edges, vertices, along axes, starting on surfaces, grazing."""
    b = scene.buffers()
    P = b["positions"].reshape(-1, 3, 3)[:36]
    eye = np.array(O.EYE, np.float32)
    targets = [P[:, 0], P[:, 1], P[:, 2], (P[:, 0] + P[:, 1]) / 2, (P[:, 1] + P[:, 2]) / 2, (P[:, 0] + P[:, 2]) / 2, P.mean(1)]
    t = np.concatenate(targets).astype(np.float32)
    n = t.shape[0]
    rays = np.zeros(n + 12, O.RAY_DTYPE)
    d = t - eye
    rays["origin"][:n] = eye
    rays["direction"][:n] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    rays["origin"][n:n + 6] = [0.0, 1.0, 0.0]
    rays["direction"][n:n + 6] = axes
    rays["origin"][n + 6:n + 12] = [[0, 0, 0], [0, 2, 0], [-1, 1, 0], [1, 1, 0], [0, 1, -1], [0.1, 0.6, 0.5]]   # on surfaces
    rays["direction"][n + 6:n + 12] = [[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 1, 0]]
    rays["mask"] = 3
    rays["maxDistance"] = np.inf
    return rays


@pytest.mark.parametrize("force_global", [0, 1])
def test_intersector(capi, O, cornell, ctx256, force_global):
    ctx256.set_option(capi.OPT_FORCE_GLOBAL, force_global)
    try:
        rays = np.concatenate([_rays(O, 60000, 21), _adversarial_rays(O, cornell)])
        ref = O.intersect_nearest(cornell, rays)
        ref_any = O.intersect_any(cornell, rays)
        # strict build: the triangle test is the oracle's arithmetic -> identical records
        ctx256.set_option(capi.OPT_STRICT, 1)
        got = ctx256.trace(rays)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), "strict nearest-hit differs from the oracle"
        got_any = ctx256.trace(rays, any_hit=True)
        assert np.array_equal(got_any >= 0, ref_any >= 0)
        # fast build: same primitive except where the double-precision margin says fp32 cannot decide
        ctx256.set_option(capi.OPT_STRICT, 0)
        fast = ctx256.trace(rays)
        diff = fast["primitiveIndex"] != ref["primitiveIndex"]
        # The bar is a SET, not a share (round 5): a ray may resolve to another primitive only where double-precision geometry says fp32
        # cannot decide -- its margin (barycentric slack of the hit and of every near miss, distance gap to the runner-up) below 1e-5.
        # About 1 % of these rays are such: the cubes stand ON the floor, so a ray that leaves a cube downwards meets the cube's bottom and
        # the floor at the same distance, and which of the two an fp32 test names is a coin toss (the strict build follows the oracle's
        # tie rule bit for bit, above).  Measured: Moeller-Trumbore 106 of the 60,264, plane form 151, box leaves 183 -- the share moves
        # with every change of the arithmetic and means nothing; that no DECIDABLE ray flips is what is asserted.
        _, _, margin = O.nearest_f64(cornell, rays)
        undecidable = margin < 1e-5
        assert not (diff & ~undecidable).any(), "a ray that double precision can decide picked another primitive: %d" % int((diff & ~undecidable).sum())
        assert 0.002 < undecidable.mean() < 0.03   # (the ray set has its ties and edge shots, and is not made of them)
        same = ~diff & (ref["primitiveIndex"] >= 0)
        np.testing.assert_allclose(fast["distance"][same], ref["distance"][same], rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose(fast["coordinates"][same], ref["coordinates"][same], rtol=0, atol=2e-5)
        fast_any = ctx256.trace(rays, any_hit=True)
        assert ((fast_any >= 0) != (ref_any >= 0)).mean() < 2e-3
        inactive = rays["maxDistance"] < 0
        assert (fast["distance"][inactive] < 0).all() and (fast["primitiveIndex"][inactive] == -1).all() and (fast_any[inactive] < 0).all()
        assert ctx256.trace(rays[:0]).shape == (0,)
    finally:
        ctx256.set_option(capi.OPT_FORCE_GLOBAL, 0)
        ctx256.set_option(capi.OPT_STRICT, 0)


def test_intersector_large_scene_in_hbm(capi, O):
    """A scene too big for LDS (2,628 triangles) takes the global-memory traversal; the oracle uses its own BVH."""
    scene = O.OracleScene.cornell_lattice(6)
    c = make_ctx(O, scene, 64, 64)
    try:
        st = c.stats()
        assert st.scene_in_lds == 0 and st.bvh_depth <= 40
        rays = _rays(O, 40000, 8, hi=(0.95, 1.9, 0.95))
        ref = O.intersect_nearest(scene, rays)
        c.set_option(capi.OPT_STRICT, 1)
        got = c.trace(rays)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        assert np.array_equal(c.trace(rays, any_hit=True) >= 0, O.intersect_any(scene, rays) >= 0)
        # and the whole path on that scene, both megakernels, against the oracle (which uses its own BVH)
        O.set_trig_mode(O.TRIG_PORTABLE)
        ref_img, rst = O.render(scene, 64, 64, 2, 3)
        O.set_trig_mode(O.TRIG_LIBM)
        for k in KERNELS:
            c.set_option(capi.OPT_KERNEL, k)
            c.reset_stats()
            c.render(0, 2, 3)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref_img)) and c.stats().rays == rst.rays
    finally:
        c.close()


def test_c4_million_triangle_scene_parity(capi, O):
    """Config C4 (Cornell box + 44^3 cubes = 1,022,244 triangles, scene in HBM, 4-wide BVH) at 256x256, 1 spp,
    3 bounces: the strict build is bit-exact against the oracle (which walks its own median-split BVH), both
    megakernels; the shipped build is within tolerance.  (parity for C4 = HIP vs oracle at 256x256.)"""
    from toyraygun_amd import host
    scene = O.OracleScene.cornell_lattice(44)
    assert scene.ntris == 1022244
    w = h = 256
    off = O.pixel_offsets(w, h)
    b = host.Scene.cornell_lattice(44).buffers()          # the product's own builder of the same scene
    c = capi.Context(w, h)
    try:
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets(off)
        st = c.stats()
        assert st.scene_in_lds == 0 and st.scene_bytes > 100e6
        O.set_trig_mode(O.TRIG_PORTABLE)
        ref, rst = O.render(scene, w, h, 1, 3, offsets=off)
        O.set_trig_mode(O.TRIG_LIBM)
        c.set_option(capi.OPT_STRICT, 1)
        for k in KERNELS:
            c.set_option(capi.OPT_KERNEL, k)
            c.reset_stats()
            c.render(0, 1, 3)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref)), "kernel %d" % k
            assert c.stats().rays == rst.rays
        ref, _ = O.render(scene, w, h, 1, 3, offsets=off)
        c.set_option(capi.OPT_STRICT, 0)
        c.set_option(capi.OPT_KERNEL, 0)
        c.render(0, 1, 3)
        rmse, frac_ok, worst = image_metrics(c.read_accum(), ref)
        assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC_C4, (rmse, frac_ok, worst)   # tests/util.py: the C4 exception and why
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


@pytest.mark.parametrize("n,builder", [(0, 1), (3, 1), (7, 1), (0, 2), (7, 2), (0, 3), (7, 3)])
def test_gpu_built_bvh_gives_identical_images(capi, O, n, builder):
    """with TRG_OPT_GPU_BUILD the acceleration structure is built on the device (binned SAH by
    levels, Karras LBVH or PLOC on the Morton order; atomic refit, emitted 4-wide).  Traversal results do not depend on the tree, so the strict
    image must still equal the oracle's bit for bit, for both megakernels; so must the intersector's records."""
    scene = O.OracleScene.cornell_lattice(n) if n else O.OracleScene.cornell_box()
    b = scene.buffers()
    w, h = 72, 56
    off = O.pixel_offsets(w, h)
    O.set_trig_mode(O.TRIG_PORTABLE)
    ref, rst = O.render(scene, w, h, 3, 4, offsets=off)
    O.set_trig_mode(O.TRIG_LIBM)
    c = capi.Context(w, h)
    try:
        c.set_option(capi.OPT_GPU_BUILD, builder)     # 1 = binned SAH by levels, 2 = Karras LBVH, 3 = PLOC merges
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        st = c.stats()
        assert st.gpu_built == 1 and st.scene_in_lds == 0 and st.bvh_nodes4 >= 1 and st.last_build_ms > 0
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets(off)
        c.set_option(capi.OPT_STRICT, 1)
        for k in KERNELS:
            c.set_option(capi.OPT_KERNEL, k)
            c.reset_stats()
            c.render(0, 3, 4)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref)) and c.stats().rays == rst.rays
        rays = np.concatenate([_rays(O, 20000, 31, hi=(0.95, 1.9, 0.95)), _adversarial_rays(O, O.OracleScene.cornell_box())])
        assert np.array_equal(c.trace(rays).view(np.uint8), O.intersect_nearest(scene, rays).view(np.uint8))
        # rebuilding with the host builder on the same context switches back
        c.set_option(capi.OPT_GPU_BUILD, 0)
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        assert c.stats().gpu_built == 0
        c.set_option(capi.OPT_KERNEL, 0)
        c.render(0, 3, 4)
        assert np.array_equal(_bits(c.read_accum()), _bits(ref))
    finally:
        c.close()


@pytest.mark.parametrize("shape", ["one_quad", "quad_and_triangle", "two_quads", "quads_and_triangles"])
@pytest.mark.parametrize("builder", [0, 1, 2, 3])
def test_smallest_scenes_with_quads(capi, O, shape, builder):
    """Quad leaves at the builders' smallest inputs (round 4): ONE quad (the device builders need two primitives and fall back to its two
    triangles), a quad and a triangle (two primitives, the root's two leaves), two quads, and a handful mixed in both index patterns -- kept in
    HBM so that the host builder (0) walks the same 4-wide path as the device builders.  Strict build: the oracle's records bit for bit; shipped
    build (one plane test per quad): the same triangles and distances away from the edges."""
    rng = np.random.default_rng(len(shape))
    s = O.OracleScene()
    eye = np.eye(4, dtype=np.float32)

    def quad(pattern):
        a = rng.uniform([-0.6, 0.4, -0.6], [0.2, 1.2, 0.2]).astype(np.float32)
        e1, e2 = rng.normal(0, 0.5, 3).astype(np.float32), rng.normal(0, 0.5, 3).astype(np.float32)
        p4 = np.stack([a, a + e1, a + e1 + e2, a + e2]).astype(np.float32)       # a, b, c, d = a + (c - b)
        s.add_geometry(p4, [0, 1, 2, 0, 2, 3] if pattern == 1 else [0, 2, 3, 0, 1, 2], eye, rng.uniform(0.2, 0.9, 3), 1)

    def tri():
        t = (rng.uniform([-0.8, 0.2, -0.8], [0.8, 1.8, 0.8], (1, 3)) + rng.normal(0, 0.4, (3, 3))).astype(np.float32)
        s.add_geometry(t, [0, 1, 2], eye, rng.uniform(0.2, 0.9, 3), 1)

    {"one_quad": lambda: quad(1), "quad_and_triangle": lambda: (quad(2), tri()), "two_quads": lambda: (quad(1), quad(2)),
     "quads_and_triangles": lambda: (tri(), quad(1), tri(), quad(2), quad(1), tri())}[shape]()
    b = s.buffers()
    rays = _rays(O, 20000, 91, lo=(-0.9, 0.1, -0.9), hi=(0.9, 1.9, 0.9))
    ref = O.intersect_nearest(s, rays, brute=True)
    assert (ref["distance"] >= 0).sum() > 200
    c = capi.Context(16, 16)
    try:
        c.set_option(capi.OPT_FORCE_GLOBAL, 1)
        c.set_option(capi.OPT_GPU_BUILD, builder)
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_option(capi.OPT_STRICT, 1)
        assert np.array_equal(c.trace(rays).view(np.uint8), ref.view(np.uint8))
        c.set_option(capi.OPT_STRICT, 0)
        got = c.trace(rays)
        same = got["primitiveIndex"] == ref["primitiveIndex"]
        assert same.mean() > 0.998          # (a ray on an edge or on a quad's diagonal may name the neighbouring triangle)
        hit = same & (ref["primitiveIndex"] >= 0)
        assert np.allclose(got["distance"][hit], ref["distance"][hit], rtol=2e-5, atol=2e-6)
        assert np.abs(got["coordinates"][hit] - ref["coordinates"][hit]).max() < 2e-4
    finally:
        c.close()


def _random_soup(O, n, seed):
    """A hostile triangle soup inside the Cornell room: random triangles plus exact duplicates (tie-break by
    primitive index), coplanar overlapping pairs, zero-area and needle triangles, a few emissive / masked ones."""
    rng = np.random.default_rng(seed)
    s = O.OracleScene.cornell_box()
    eye = np.eye(4, dtype=np.float32)
    ctr = rng.uniform([-0.8, 0.2, -0.8], [0.8, 1.8, 0.8], (n, 3)).astype(np.float32)
    tri = ctr[:, None, :] + rng.normal(0, 0.12, (n, 3, 3)).astype(np.float32)
    tri[n // 2: n // 2 + n // 10] = tri[: n // 10]                       # exact duplicates
    tri[n // 2 + n // 10: n // 2 + n // 5, :, 1] = np.float32(1.0)        # coplanar, overlapping (y = 1 plane)
    tri[-8:-4, 2] = tri[-8:-4, 1]                                          # zero-area (two equal vertices)
    tri[-4:, 2] = tri[-4:, 0] + (tri[-4:, 1] - tri[-4:, 0]) * np.float32(0.5)   # zero-area (collinear)
    mats = rng.choice([1, 1, 1, 1, 2, 3], n).astype(np.uint32)
    for k in range(n):
        s.add_geometry(tri[k], [0, 1, 2], eye, rng.uniform(0.2, 0.9, 3), int(mats[k]))
    return s


@pytest.mark.parametrize("ntris", [2, 3, 4])
@pytest.mark.parametrize("builder", [1, 2, 3])
def test_gpu_build_of_a_handful_of_triangles(capi, O, ntris, builder):
    """The device builders' smallest inputs (a root with two leaves; an odd triangle out): records identical to brute force."""
    rng = np.random.default_rng(ntris)
    s = O.OracleScene()
    eye = np.eye(4, dtype=np.float32)
    for k in range(ntris):
        tri = (rng.uniform([-0.8, 0.2, -0.8], [0.8, 1.8, 0.8], (1, 3)) + rng.normal(0, 0.4, (3, 3))).astype(np.float32)
        s.add_geometry(tri, [0, 1, 2], eye, (0.5, 0.5, 0.5), 1)
    b = s.buffers()
    c = capi.Context(16, 16)
    try:
        c.set_option(capi.OPT_GPU_BUILD, builder)
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        assert c.stats().gpu_built == 1
        c.set_option(capi.OPT_STRICT, 1)
        rays = _rays(O, 20000, 77 + ntris, hi=(0.95, 1.9, 0.95))
        ref = O.intersect_nearest(s, rays, brute=True)
        assert np.array_equal(c.trace(rays).view(np.uint8), ref.view(np.uint8))
        assert (ref["distance"] >= 0).sum() > 100
    finally:
        c.close()


@pytest.mark.parametrize("scale,shift,gpu_build", [(100.0, (5000.0, -3000.0, 800.0), 0), (0.01, (-7.0, 3.0, 11.0), 0),
                                                   (100.0, (5000.0, -3000.0, 800.0), 1), (100.0, (5000.0, -3000.0, 800.0), 2),
                                                   (100.0, (5000.0, -3000.0, 800.0), 3)])
def test_intersector_far_from_the_origin(capi, O, scale, shift, gpu_build):
    """Quantised wide nodes (8-bit child boxes against the node box) and the LBVH on geometry that is large / tiny and far
    from the origin: nearest and any-hit queries stay bit-identical to the oracle, whose own tree uses float boxes."""
    rng = np.random.default_rng(77)
    n = 2500
    ctr = rng.uniform(-1, 1, (n, 3))
    tri = (ctr[:, None, :] + rng.normal(0, 0.05, (n, 3, 3))) * scale + np.asarray(shift)
    tri = tri.astype(np.float32)
    s = O.OracleScene()
    eye = np.eye(4, dtype=np.float32)
    for k in range(n):
        s.add_geometry(tri[k], [0, 1, 2], eye, (0.5, 0.5, 0.5), 1)
    b = s.buffers()
    rays = np.zeros(30000, O.RAY_DTYPE)
    rays["origin"] = (rng.uniform(-1.2, 1.2, (30000, 3)) * scale + np.asarray(shift)).astype(np.float32)
    d = rng.normal(size=(30000, 3))
    d[:300, 0] = 0.0; d[300:600, 1] = 0.0; d[600:700, :2] = 0.0     # axis-parallel rays: zero direction components
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["mask"] = 1
    rays["maxDistance"] = np.where(rng.random(30000) < 0.3, rng.uniform(0.05, 2.0, 30000) * scale, np.inf).astype(np.float32)
    ref = O.intersect_nearest(s, rays)
    assert (ref["distance"] >= 0).mean() > 0.2
    c = capi.Context(16, 16)
    try:
        c.set_option(capi.OPT_GPU_BUILD, gpu_build)
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        assert c.stats().scene_in_lds == 0
        for strict in (1, 0):
            c.set_option(capi.OPT_STRICT, strict)
            got = c.trace(rays)
            if strict:
                assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
            else:  # shipped build: same primitive on all but edge-grazing rays, distances within rounding
                same = got["primitiveIndex"] == ref["primitiveIndex"]
                assert same.mean() > 0.999
                hit = same & (ref["distance"] >= 0)
                assert np.allclose(got["distance"][hit], ref["distance"][hit], rtol=1e-4, atol=1e-6 * scale)
            got_any, ref_any = c.trace(rays, any_hit=True) >= 0, O.intersect_any(s, rays) >= 0
            if strict:
                assert np.array_equal(got_any, ref_any)
            else:   # shipped build: an occlusion answer may flip only where a ray grazes an edge (bounded like test_intersector's any-hit leg)
                assert (got_any != ref_any).mean() < 2e-3, (got_any != ref_any).mean()
    finally:
        c.close()


@pytest.mark.parametrize("n,seed", [(40, 11), (90, 12)])
def test_lds_resident_soup_parity(capi, O, n, seed):
    """The same hostile soup small enough to be staged in LDS (sign-ordered BVH2 nodes): the intersector and the whole
    path, strict build bit-exact against the oracle, the LDS-resident and the forced-HBM (quantised 4-wide) traversal of
    the SAME scene agreeing with it and with each other, frame split and tiny / ragged image sizes included."""
    scene = _random_soup(O, n, seed)
    b = scene.buffers()
    rays = _rays(O, 20000, 300 + seed, hi=(0.95, 1.9, 0.95))
    ref_hits = O.intersect_nearest(scene, rays, brute=True)
    ref_any = O.intersect_any(scene, rays) >= 0
    for (w, h) in ((64, 48), (1, 1), (9, 1), (3, 19)):
        off = O.pixel_offsets(w, h)
        O.set_trig_mode(O.TRIG_PORTABLE)
        try:
            img_ref, rst = O.render(scene, w, h, 5, 4, offsets=off)
        finally:
            O.set_trig_mode(O.TRIG_LIBM)
        c = capi.Context(w, h)
        try:
            c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
            c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
            c.set_pixel_offsets(off)
            c.set_option(capi.OPT_STRICT, 1)
            for force_global in (0, 1):
                c.set_option(capi.OPT_FORCE_GLOBAL, force_global)
                if (w, h) == (64, 48):
                    assert np.array_equal(c.trace(rays).view(np.uint8), ref_hits.view(np.uint8))
                    assert np.array_equal(c.trace(rays, any_hit=True) >= 0, ref_any)
                for fsplit in (1, 2, 4):
                    c.set_option(capi.OPT_FRAME_SPLIT, fsplit)
                    c.reset_stats()
                    c.render(0, 5, 4)
                    st = c.stats()
                    assert np.array_equal(_bits(c.read_accum()), _bits(img_ref)), (w, h, force_global, fsplit)
                    assert st.rays == rst.rays and st.scene_in_lds == (0 if force_global else 1)
        finally:
            c.close()


def _box_rays(O, scene, boxes, seed):
    """Random rays in the room, rays that START INSIDE every box (the triangles are two-sided: they meet the far face from within), and rays from
    the eye and from random points aimed at every box's corners, edge midpoints and face centres."""
    rng = np.random.default_rng(seed)
    parts = [_rays(O, 30000, seed, hi=(0.95, 1.9, 1.5))]
    eye = np.array(O.EYE, np.float32)
    for row in boxes:
        c, A = row[2:5].astype(np.float64), row[5:14].reshape(3, 3).astype(np.float64)
        H = np.linalg.inv(A)                                            # columns: the half axes
        inside = _rays(O, 600, seed + 1)
        inside["origin"] = (c + (H @ rng.uniform(-0.95, 0.95, (3, 600))).T).astype(np.float32)
        grid = np.array([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if (x, y, z) != (0, 0, 0)], np.float64)
        targets = (c + (H @ grid.T).T).astype(np.float32)               # 8 corners, 12 edge midpoints, 6 face centres
        aimed = np.zeros(2 * len(targets), O.RAY_DTYPE)
        org = np.concatenate([np.tile(eye, (len(targets), 1)), rng.uniform((-0.9, 0.1, -0.9), (0.9, 1.9, 2.5), (len(targets), 3)).astype(np.float32)])
        d = np.concatenate([targets, targets]) - org
        aimed["origin"], aimed["direction"] = org, (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        aimed["mask"], aimed["maxDistance"] = 3, np.inf
        parts += [inside, aimed]
    return np.concatenate(parts)


def test_box_leaves(capi, O, monkeypatch):
    """Round 5: addCube's twelve triangles as ONE leaf of a scene staged in LDS (bvh_build.h kLeafBox; tests/util.py box_zoo: rotated, sheared,
    mirrored, nested, emissive, material-3 cubes and an almost-cube).  The strict build tests the twelve triangles of such a leaf and stays
    bit-identical to the oracle -- records, any-hit answers, the image with its ray counts --; the shipped build does one slab test in the
    box's own frame and may differ from the oracle only where double precision says fp32 cannot decide (rays that graze an edge or meet
    two coincident faces), hit distances and weights to 3e-6 / 2e-5 elsewhere, the image within the stated tolerance.  Switched off
    (TRG_BVH_BOXES=0) the same scene gives the same strict bits and clearly more primitive tests per ray: the leaves are really in use."""
    from tests.util import box_zoo
    scene, n_boxes = box_zoo(O)
    b = scene.buffers()
    boxes = capi.debug_boxes(b["positions"], b["indices"], b["material_ids"])
    assert boxes.shape[0] == n_boxes
    rays = _box_rays(O, scene, boxes, 55)
    ref = O.intersect_nearest(scene, rays, brute=True)
    ref_any = O.intersect_any(scene, rays) >= 0
    _, _, margin = O.nearest_f64(scene, rays)
    undecidable = margin < 1e-5
    assert undecidable.mean() < 0.05 and (ref["primitiveIndex"] >= 36).mean() > 0.05       # the extra boxes are hit by thousands of these rays, and decidably
    w, h, spp, bnc = 64, 48, 6, 4
    off = O.pixel_offsets(w, h)
    O.set_trig_mode(O.TRIG_PORTABLE)
    try:
        img_ref, rst = O.render(scene, w, h, spp, bnc, offsets=off)
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
    img_libm, _ = O.render(scene, w, h, spp, bnc, offsets=off)
    tests = {}
    for switch in ("1", "0"):
        monkeypatch.setenv("TRG_BVH_BOXES", switch)
        c = capi.Context(w, h)
        try:
            c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
            c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
            c.set_pixel_offsets(off)
            assert c.stats().scene_in_lds == 1 and c.stats().bvh_boxes == (n_boxes if switch == "1" else 0)
            for force_global in (0, 1):      # the tree staged in LDS, and the same scene traversed from HBM (the shipped build has box leaves in both)
                c.set_option(capi.OPT_FORCE_GLOBAL, force_global)
                tag = (switch, force_global)
                c.set_option(capi.OPT_STRICT, 1)
                assert np.array_equal(c.trace(rays).view(np.uint8), ref.view(np.uint8)), "strict nearest-hit differs from the oracle %s" % (tag,)
                assert np.array_equal(c.trace(rays, any_hit=True) >= 0, ref_any)
                c.reset_stats()
                c.render(0, spp, bnc)
                assert np.array_equal(_bits(c.read_accum()), _bits(img_ref)) and c.stats().rays == rst.rays
                c.set_option(capi.OPT_STRICT, 0)
                fast = c.trace(rays)
                diff = fast["primitiveIndex"] != ref["primitiveIndex"]
                assert not (diff & ~undecidable).any(), "%s: %d decidable rays picked another primitive" % (tag, int((diff & ~undecidable).sum()))
                same = ~diff & (ref["primitiveIndex"] >= 0)
                np.testing.assert_allclose(fast["distance"][same], ref["distance"][same], rtol=3e-6, atol=3e-6)
                # the weights say WHERE on the triangle the hit lies: the point they name must be the oracle's to 1e-5 of the room's size, and the
                # weights themselves to 2e-5 on a triangle of the room's size -- 1e-5 / (shortest edge) on the zoo's 10 cm cubes, where the same
                # few ulps of position are a larger share of the triangle
                T = b["positions"].reshape(-1, 3, 3)[ref["primitiveIndex"][same]].astype(np.float64)
                uvf, uvr = fast["coordinates"][same].astype(np.float64), ref["coordinates"][same].astype(np.float64)
                point = lambda uv: T[:, 0] + uv[:, :1] * (T[:, 1] - T[:, 0]) + uv[:, 1:] * (T[:, 2] - T[:, 0])
                assert np.abs(point(uvf) - point(uvr)).max() < 1e-5, tag
                edge = np.minimum(np.linalg.norm(T[:, 1] - T[:, 0], axis=1), np.linalg.norm(T[:, 2] - T[:, 0], axis=1))
                assert (np.abs(uvf - uvr).max(1) <= np.maximum(2e-5, 1e-5 / edge)).all(), tag
                fast_any = c.trace(rays, any_hit=True) >= 0
                assert (fast_any != ref_any).mean() < 2e-3
                for regen in ((-1,) if not force_global else (0, 1)):      # (HBM: the lock-step megakernel and path regeneration)
                    c.set_option(capi.OPT_REGEN, regen)
                    c.set_option(capi.OPT_COUNTERS, 1)
                    c.reset_stats()
                    c.render(0, spp, bnc)
                    st = c.stats()
                    tests[tag] = (st.tri_tests / st.rays, st.node_fetches / st.rays)
                    rmse, frac_ok, _ = image_metrics(c.read_accum(), img_libm)
                    assert rmse <= TOL_RMSE and frac_ok >= 1.0 - edge_flip_allowance(w * h, st.rays) / (w * h), (tag, regen, rmse, frac_ok)
                    c.set_option(capi.OPT_COUNTERS, 0)
                c.set_option(capi.OPT_REGEN, -1)
        finally:
            c.close()
    # the box leaves are in use: fewer primitive tests per ray (one per box where a cube took one or two quads) and fewer node steps (no subtree)
    for fg in (0, 1):
        assert tests[("1", fg)][0] < tests[("0", fg)][0] and tests[("1", fg)][1] < 0.97 * tests[("0", fg)][1], tests


@pytest.mark.parametrize("n,seed,gpu_build", [(300, 1, 0), (300, 2, 1), (3000, 3, 0), (3000, 4, 1), (3000, 5, 2), (3000, 6, 3), (2, 7, 1), (5, 8, 1)])
def test_random_triangle_soup_parity(capi, O, n, seed, gpu_build):
    """Fuzz: the intersector and the whole path on random, partly degenerate geometry -- strict build bit-exact against
    the oracle's brute force / BVH, host-built and GPU-built trees, both megakernels."""
    scene = _random_soup(O, n, seed)
    b = scene.buffers()
    w, h = 64, 48
    off = O.pixel_offsets(w, h)
    c = capi.Context(w, h)
    try:
        c.set_option(capi.OPT_GPU_BUILD, gpu_build)
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets(off)
        c.set_option(capi.OPT_STRICT, 1)
        rays = _rays(O, 30000, 100 + seed, hi=(0.95, 1.9, 0.95))
        ref = O.intersect_nearest(scene, rays, brute=(n <= 300))
        got = c.trace(rays)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        assert np.array_equal(c.trace(rays, any_hit=True) >= 0, O.intersect_any(scene, rays) >= 0)
        O.set_trig_mode(O.TRIG_PORTABLE)
        img_ref, rst = O.render(scene, w, h, 2, 4, offsets=off)
        O.set_trig_mode(O.TRIG_LIBM)
        for k in KERNELS:
            c.set_option(capi.OPT_KERNEL, k)
            c.reset_stats()
            c.render(0, 2, 4)
            assert np.array_equal(_bits(c.read_accum()), _bits(img_ref)) and c.stats().rays == rst.rays
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


def test_degenerate_scenes(capi, O):
    """Empty scene and a single triangle (the BVH builder synthesises the root)."""
    u = O.make_uniforms(32, 32)
    c = capi.Context(32, 32)
    try:
        c.set_uniforms(O.uniforms_bytes(u))
        c.set_pixel_offsets_seed()
        z3, zi = np.zeros((0, 3), np.float32), np.zeros(0, np.uint32)
        c.load_scene(z3, z3, z3, zi, zi)
        c.render(0, 2, 3)
        img = c.read_accum()
        assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()
        st = c.stats()
        assert st.primary_rays == 32 * 32 * 2 and st.shadow_rays == 0 and st.bounce_rays == 0
        # one big emissive triangle in front of the camera: hit pixels are exactly the light colour
        tri = np.array([[-5, -5, 0], [5, -5, 0], [0, 8, 0]], np.float32)
        nrm = np.tile(np.array([[0, 0, 1]], np.float32), (3, 1))
        c.load_scene(tri, nrm, np.ones((3, 3), np.float32), [0, 1, 2], [2])
        c.render(0, 1, 3)
        img = c.read_accum()
        assert (img[..., :3] == 1.0).all()
    finally:
        c.close()


# ------------------------------------------------------------------ whole path
# The schedules of the same arithmetic the loaded library has: TRG_KERNEL_DIRECT (0) in the product library; with
# TRG_HIP_SO=experiments/lib/libtoyraygun_hip_exp.so also TRG_KERNEL_POOL (1) and TRG_KERNEL_WAVEFRONT (2) -- the schedules that lost and live
# in experiments/ -- and the whole suite then runs all three (tests/test_experiments.py is their smoke test in the default run).
def _kernels():
    from toyraygun_amd import capi as _capi
    try:
        return [0, 1, 2] if _capi.has_experiments() else [0]
    except (ImportError, OSError):
        return [0]


KERNELS = _kernels()


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("w,h,spp,bounces", [(256, 256, 1, 1), (96, 64, 4, 3), (48, 32, 2, 8), (33, 17, 3, 15), (40, 40, 5, 2)])
def test_render_strict_is_bit_exact(capi, O, cornell, w, h, spp, bounces, kernel):
    off = O.pixel_offsets(w, h)
    c = make_ctx(O, cornell, w, h, offsets=off)
    try:
        O.set_trig_mode(O.TRIG_PORTABLE)
        ref, st = O.render(cornell, w, h, spp, bounces, offsets=off)
        c.set_option(capi.OPT_STRICT, 1)
        c.set_option(capi.OPT_KERNEL, kernel)
        c.render(0, spp, bounces)
        img, gs = c.read_accum(), c.stats()
        assert np.array_equal(_bits(img), _bits(ref))
        assert (gs.primary_rays, gs.bounce_rays, gs.shadow_rays, gs.shaded_hits) == (st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits)
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


def test_render_c1_fast_within_tolerance_and_golden(capi, O, cornell, ctx256, golden):
    """Config C1: 256x256, 1 spp, 1 bounce."""
    off = O.pixel_offsets(256, 256)
    ref, st = O.render(cornell, 256, 256, 1, 1, offsets=off)
    ctx256.set_option(capi.OPT_STRICT, 0)
    ctx256.reset_stats()
    ctx256.render(0, 1, 1)
    img, gs = ctx256.read_accum(), ctx256.stats()
    rmse, frac_ok, worst = image_metrics(img, ref)
    assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
    rmse, frac_ok, _ = image_metrics(img, golden["frame_c1_256_1spp_1b"])
    assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC
    assert gs.rays == st.rays == int(golden["rays_c1"][:3].sum())
    assert (img[..., 3] == 1.0).all()


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("w,h,spp,bounces", [(128, 96, 16, 3), (64, 64, 8, 8)])
def test_render_fast_within_tolerance(capi, O, cornell, w, h, spp, bounces, kernel):
    off = O.pixel_offsets(w, h)
    c = make_ctx(O, cornell, w, h, offsets=off)
    try:
        ref, st = O.render(cornell, w, h, spp, bounces, offsets=off)
        c.set_option(capi.OPT_KERNEL, kernel)
        c.render(0, spp, bounces)
        img, gs = c.read_accum(), c.stats()
        rmse, frac_ok, worst = image_metrics(img, ref)
        assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
        assert abs(gs.rays - st.rays) <= 1e-4 * st.rays
    finally:
        c.close()


def test_golden_frames_on_gpu(capi, O, cornell, golden):
    c = make_ctx(O, cornell, 64, 64)
    try:
        c.set_option(capi.OPT_STRICT, 1)
        c.render(0, 4, 3)
        d = np.abs(c.read_accum() - golden["frame_64_4spp_3b_portable"])
        assert (d.max(-1) > 1e-5).mean() < 0.002
        assert c.stats().rays == int(golden["rays_64_4spp_3b_portable"][:3].sum())
    finally:
        c.close()


@pytest.mark.parametrize("fsplit", [2, 4])
@pytest.mark.parametrize("force_global", [0, 1])
def test_frame_split_schedule_is_bit_exact(capi, O, cornell, fsplit, force_global):
    """TRG_OPT_FRAME_SPLIT: 2 or 4 wavefronts share a pixel's frames and one of them folds the parked radiances in
    frame order -- bit-identical to the oracle in the strict build (spp not a multiple of the chunk, more frames than
    one chunk parks, ragged image sizes, continuation, row bands), within tolerance in the fast build."""
    for (w, h, spp, bounces) in ((96, 64, 4, 3), (33, 17, 7, 5), (40, 40, 19, 2), (64, 8, 2, 3)):
        off = O.pixel_offsets(w, h)
        c = make_ctx(O, cornell, w, h, offsets=off)
        try:
            O.set_trig_mode(O.TRIG_PORTABLE)
            ref, st = O.render(cornell, w, h, spp, bounces, offsets=off)
            c.set_option(capi.OPT_FORCE_GLOBAL, force_global)
            c.set_option(capi.OPT_FRAME_SPLIT, fsplit)
            c.set_option(capi.OPT_STRICT, 1)
            c.reset_stats()
            c.render(0, spp, bounces)
            img, gs = c.read_accum(), c.stats()
            assert np.array_equal(_bits(img), _bits(ref)), (w, h, spp, bounces)
            assert (gs.primary_rays, gs.bounce_rays, gs.shadow_rays, gs.shaded_hits) == (st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits)
            # continuation + uneven row bands, mixed with the frame-serial kernel
            c.render(0, 1, bounces)
            c.set_option(capi.OPT_FRAME_SPLIT, 1)
            c.render(1, 1, bounces)
            c.set_option(capi.OPT_FRAME_SPLIT, fsplit)
            if spp > 2:
                for row0, rows in ((0, 3), (3, h - 4), (h - 1, 1)):
                    c.render(2, spp - 2, bounces, row0, rows)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref))
            # counters instantiation
            c.set_option(capi.OPT_COUNTERS, 1)
            c.reset_stats()
            c.render(0, spp, bounces)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref))
            assert c.stats().node_fetches > 0
            c.set_option(capi.OPT_COUNTERS, 0)
            # fast build: tolerance against the faithful (libm) oracle
            O.set_trig_mode(O.TRIG_LIBM)
            ref_l, _ = O.render(cornell, w, h, spp, bounces, offsets=off)
            c.set_option(capi.OPT_STRICT, 0)
            c.render(0, spp, bounces)
            rmse, frac_ok, _ = image_metrics(c.read_accum(), ref_l)
            assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok)
        finally:
            O.set_trig_mode(O.TRIG_LIBM)
            c.close()


def test_frame_split_auto_selection(capi, O, cornell):
    """Auto mode (default) picks the frame-parallel schedule for small grids; whatever it picks, the strict result is
    the frame-serial one bit for bit, and bad values are rejected."""
    w, h = 320, 200
    c = make_ctx(O, cornell, w, h)
    try:
        c.set_option(capi.OPT_STRICT, 1)
        c.set_option(capi.OPT_FRAME_SPLIT, 1)
        c.render(0, 16, 3)
        serial = c.read_accum()
        c.set_option(capi.OPT_FRAME_SPLIT, 0)
        c.render(0, 16, 3)
        assert np.array_equal(_bits(serial), _bits(c.read_accum()))
        with pytest.raises(capi.TrgError):
            c.set_option(capi.OPT_FRAME_SPLIT, 3)
    finally:
        c.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_continuation_bands_and_determinism(capi, O, cornell, kernel):
    """Size-independent properties: frames [0,6) == [0,2)+[2,6); union of row bands == full frame;
    repeated launches are bitwise identical; LDS-resident and HBM-resident scenes agree bitwise."""
    w, h = 80, 48
    c = make_ctx(O, cornell, w, h)
    try:
        c.set_option(capi.OPT_KERNEL, kernel)
        for strict in (0, 1):
            c.set_option(capi.OPT_STRICT, strict)
            c.render(0, 6, 3)
            full = c.read_accum()
            c.render(0, 6, 3)
            assert np.array_equal(_bits(full), _bits(c.read_accum()))
            c.render(0, 2, 3)
            c.render(2, 4, 3)
            assert np.array_equal(_bits(full), _bits(c.read_accum()))
            c.render(0, 1, 3)
            c.render(1, 5, 3)            # odd split: the pool kernel batches frames in pairs
            assert np.array_equal(_bits(full), _bits(c.read_accum()))
            c.render(0, 1, 3)            # clobber, then rebuild from uneven bands
            for row0, rows in ((0, 7), (7, 16), (23, 1), (24, 24)):
                c.render(0, 6, 3, row0, rows)
            assert np.array_equal(_bits(full), _bits(c.read_accum()))
            # other kernel instantiations (scene in HBM, counters on): bitwise equal in the strict build;
            # in the fast build the compiler may contract differently per instantiation -> tolerance
            def same(img):
                if strict:
                    return np.array_equal(_bits(full), _bits(img))
                rmse, frac_ok, _ = image_metrics(img, full)
                return rmse <= TOL_RMSE and frac_ok >= TOL_FRAC
            c.set_option(capi.OPT_FORCE_GLOBAL, 1)
            c.render(0, 6, 3)
            assert same(c.read_accum())
            c.set_option(capi.OPT_FORCE_GLOBAL, 0)
            c.set_option(capi.OPT_COUNTERS, 1)
            c.reset_stats()
            c.render(0, 6, 3)
            st = c.stats()
            assert same(c.read_accum())
            assert st.node_fetches > st.rays and st.tri_tests > 0
            c.set_option(capi.OPT_COUNTERS, 0)
    finally:
        c.close()


def _rows_against_oracle(capi, O, c, scene, w, h, spp, bnc, img, rows, frac=TOL_FRAC):
    """The full-size frame `img` of the shipped build on sampled row bands [(row0, n), ...] against the oracle within the stated
    tolerance; then the same frame from the strict build on the same rows bit for bit.  (The oracle renders bands independently; the
    whole frame would take it about a minute per config on the box's host threads.)"""
    off = O.pixel_offsets(w, h)
    acc = np.zeros((h, w, 4), np.float32)
    for r0, n in rows:
        O.render(scene, w, h, spp, bnc, row0=r0, rows=n, accum=acc, offsets=off)
    got = np.concatenate([img[r0:r0 + n] for r0, n in rows])
    ref = np.concatenate([acc[r0:r0 + n] for r0, n in rows])
    rmse, frac_ok, worst = image_metrics(got, ref)
    assert rmse <= TOL_RMSE and frac_ok >= frac, (rmse, frac_ok, worst)
    O.set_trig_mode(O.TRIG_PORTABLE)
    try:
        acc[:] = 0
        for r0, n in rows:
            O.render(scene, w, h, spp, bnc, row0=r0, rows=n, accum=acc, offsets=off)
        c.set_option(capi.OPT_STRICT, 1)
        c.render(0, spp, bnc)
        strict = c.read_accum()
        for r0, n in rows:
            assert np.array_equal(_bits(strict[r0:r0 + n]), _bits(acc[r0:r0 + n])), (r0, n)
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.set_option(capi.OPT_STRICT, 0)


def test_full_size_c2_properties(capi, O, cornell):
    """Config C2 (1920x1080, 16 spp, 3 bounces) at full size: invariants of the whole frame, and the WHOLE frame against the oracle
    (all 1080 rows, 33 M paths; the oracle on the box's host threads needs a few seconds): the shipped build within the stated
    tolerance, the strict build bit for bit with the oracle's ray counts."""
    w, h, spp, bnc = 1920, 1080, 16, 3
    c = make_ctx(O, cornell, w, h)
    try:
        c.render(0, spp, bnc)
        img, st = c.read_accum(), c.stats()
        assert np.isfinite(img).all() and (img[..., 3] == 1.0).all() and (img[..., :3] >= 0).all()
        assert st.primary_rays == w * h * spp and st.shadow_rays == st.shaded_hits
        assert st.bounce_rays <= st.shaded_hits and st.rays <= 2 * bnc * w * h * spp
        # the side bars (outside the box opening: |x_ndc| > ~0.6 at 16:9) never hit anything
        assert (img[:, :300, :3] == 0).all() and (img[:, -300:, :3] == 0).all()
        # left wall red, right wall green (row 0 = scene bottom)
        left, right = img[400:700, 480:560, :3].mean((0, 1)), img[400:700, 1360:1440, :3].mean((0, 1))
        assert left[0] > 3 * left[1] and right[1] > 2 * right[0]
        # light texels are exactly the light colour and sit in the upper half
        ys, _ = np.where((img[..., :3] == 1.0).all(-1))
        assert ys.size > 1000 and ys.min() > h // 2
        off = O.pixel_offsets(w, h)
        ref, _ = O.render(cornell, w, h, spp, bnc, offsets=off)
        rmse, frac_ok, worst = image_metrics(img, ref)
        assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
        O.set_trig_mode(O.TRIG_PORTABLE)
        ref, rst = O.render(cornell, w, h, spp, bnc, offsets=off)
        c.set_option(capi.OPT_STRICT, 1)
        c.reset_stats()
        c.render(0, spp, bnc)
        assert np.array_equal(_bits(c.read_accum()), _bits(ref))
        gs = c.stats()
        assert (gs.primary_rays, gs.bounce_rays, gs.shadow_rays) == (rst.primary_rays, rst.bounce_rays, rst.shadow_rays)
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


def test_full_size_c3_and_c5_properties(capi, O, cornell):
    """Configs C3 (1920x1080, 256 spp, 8 bounces) and C5 (3840x2160, 64 spp, 3 bounces; here on one GPU, sharded into the
    eight row bands of the 8-GPU job) at BASELINE's full sizes, through size-independent properties: a launch split in
    two continues bit for bit, the union of the bands is the unsharded frame bit for bit, ray-count identities hold,
    and sampled rows agree with the oracle within the stated tolerance."""
    from toyraygun_amd.dist import band_rows
    # ---- C3
    w, h, spp, bnc = 1920, 1080, 256, 8
    c = make_ctx(O, cornell, w, h)
    try:
        c.reset_stats()
        c.render(0, spp, bnc)
        img, st = c.read_accum(), c.stats()
        assert np.isfinite(img).all() and (img[..., 3] == 1.0).all() and (img[..., :3] >= 0).all()
        assert st.primary_rays == w * h * spp and st.shadow_rays == st.shaded_hits and st.bounce_rays <= st.shaded_hits
        assert st.rays <= 2 * bnc * w * h * spp and st.bounce_rays > st.primary_rays   # deep bounces really happen
        c.render(0, 100, bnc)
        c.render(100, spp - 100, bnc)
        assert np.array_equal(_bits(c.read_accum()), _bits(img))
        _rows_against_oracle(capi, O, c, cornell, w, h, spp, bnc, img, [(300, 2), (700, 2), (1040, 2)])
    finally:
        c.close()
    # ---- C5
    w, h, spp, bnc = 3840, 2160, 64, 3
    c = make_ctx(O, cornell, w, h)
    try:
        c.reset_stats()
        c.render(0, spp, bnc)
        full, st = c.read_accum(), c.stats()
        assert st.primary_rays == w * h * spp and st.shadow_rays == st.shaded_hits
        c.render(0, 1, bnc)                                   # clobber, then the eight bands of the 8-GPU job
        c.reset_stats()
        for r in range(8):
            row0, rows = band_rows(h, 8, r)
            c.render(0, spp, bnc, row0, rows)
        assert np.array_equal(_bits(c.read_accum()), _bits(full)) and c.stats().rays == st.rays
        # two rows across every boundary between the eight bands, and two more near the bottom and the top
        _rows_against_oracle(capi, O, c, cornell, w, h, spp, bnc, full, [(100, 2)] + [(270 * k - 1, 2) for k in range(1, 8)] + [(2050, 2)])
        # the same eight ranks with INTERLEAVED bands (round 4: the default of the 8-GPU job -- 8-row micro-bands dealt round robin, rendered
        # compactly, unpacked after the exchange): the union is the unsharded frame bit for bit, and the ranks' ray counts differ by under 1.5 %
        import torch
        rows0, stride = capi.microband_rows(h, 8, 0)
        compact = torch.zeros((8 * stride, w, 4), dtype=torch.float32, device="cuda")
        image = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        c.bind_accum(compact.data_ptr())
        per_rank = []
        for r in range(8):
            c.reset_stats()
            c.render_bands(0, spp, bnc, 8, r, r * stride)
            per_rank.append(c.stats().rays)
        c.unpack_bands(compact.data_ptr(), image.data_ptr(), 8)
        c.sync()
        c.bind_accum(None)
        assert np.array_equal(_bits(image.cpu().numpy()), _bits(full)) and sum(per_rank) == st.rays
        assert max(per_rank) <= 1.015 * (sum(per_rank) / 8), per_rank
    finally:
        c.close()


def test_kernels_agree(capi, O, cornell):
    c = make_ctx(O, cornell, 200, 120)
    try:
        for strict in (1, 0):
            c.set_option(capi.OPT_STRICT, strict)
            imgs, rays = [], []
            for k in KERNELS:
                c.set_option(capi.OPT_KERNEL, k)
                c.reset_stats()
                c.render(0, 6, 5)
                imgs.append(c.read_accum())
                st = c.stats()
                rays.append((st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits))
            for k in range(1, len(imgs)):
                if strict:
                    assert np.array_equal(_bits(imgs[0]), _bits(imgs[k])) and rays[0] == rays[k]
                else:
                    rmse, frac_ok, _ = image_metrics(imgs[k], imgs[0])
                    assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC
        with pytest.raises(capi.TrgError):
            c.set_option(capi.OPT_KERNEL, 7)
        if not capi.has_experiments():     # the product library refuses the two experimental schedules, and says where they live
            for k in (capi.KERNEL_POOL, capi.KERNEL_WAVEFRONT):
                with pytest.raises(capi.TrgError, match="experiments"):
                    c.set_option(capi.OPT_KERNEL, k)
    finally:
        c.close()


def test_invalid_material_is_magenta(capi, O):
    s = O.OracleScene()
    m = np.eye(4, dtype=np.float32)
    quad = np.array([[-3, -3, 0], [3, -3, 0], [3, 3, 0], [-3, 3, 0]], np.float32) + np.array([0, 1, 0], np.float32)
    s.add_geometry(quad, [0, 1, 2, 0, 2, 3], m, (0.5, 0.5, 0.5), 3)   # neither DEFAULT nor EMISSIVE
    c = make_ctx(O, s, 32, 32)
    try:
        for strict, mode in ((1, O.TRIG_PORTABLE), (0, O.TRIG_LIBM)):
            O.set_trig_mode(mode)
            ref, _ = O.render(s, 32, 32, 2, 3, offsets=O.pixel_offsets(32, 32))
            c.set_option(capi.OPT_STRICT, strict)
            for k in KERNELS:
                c.set_option(capi.OPT_KERNEL, k)
                c.render(0, 2, 3)
                img = c.read_accum()
                assert np.array_equal(_bits(img), _bits(ref))
        assert (img[..., :3] == np.array([1, 0, 1], np.float32)).all()
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


def test_error_behaviour(capi, O, cornell):
    c = capi.Context(64, 48)
    try:
        with pytest.raises(capi.TrgError) as e:
            c.render(0, 1, 3)
        assert e.value.code == capi.ERR_INVALID and "scene" in str(e.value)
        b = cornell.buffers()
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        with pytest.raises(capi.TrgError):
            c.render(0, 1, 3)                      # no uniforms yet
        with pytest.raises(capi.TrgError):
            c.set_uniforms(O.uniforms_bytes(O.make_uniforms(32, 32)))   # wrong size
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(64, 48)))
        with pytest.raises(capi.TrgError):
            c.render(0, 1, 3)                      # no pixel offsets yet
        c.set_pixel_offsets_seed()
        for args in ((0, 1, 16), (0, 1, 3, 40, 9), (0, 1, 3, 49, 0), (2 ** 32 - 1, 2, 3)):
            with pytest.raises(capi.TrgError):
                c.render(*args)
        c.render(0, 0, 3)                          # zero samples / rows: a no-op, not an error
        c.render(0, 1, 3, 10, 0)
        bad = b["indices"].copy()
        bad[5] = 1000
        with pytest.raises(capi.TrgError):
            c.load_scene(b["positions"], b["normals"], b["colors"], bad, b["material_ids"])
        with pytest.raises(capi.TrgError):
            capi.Context(16, 16, device=99)
    finally:
        c.close()


# ------------------------------------------------------------------ plugin surface + N1
def test_reference_app_call_sequence(capi, O, cornell, tmp_path):
    """Engine -> shaders -> renderer.init -> camera -> loadScene -> renderFrame loop (main.cpp:21-95): the
    progressive 1-sample-per-renderFrame loop equals one batched launch bit for bit, and the oracle within tolerance."""
    from toyraygun_amd import host
    w, h, frames = 160, 120, 8
    png = str(tmp_path / "app.png")
    prog, _, rays_p = host.run_app(w, h, frames, 3, batch=False, png_path=png)
    batch, _, rays_b = host.run_app(w, h, frames, 3, batch=True)
    assert np.array_equal(_bits(prog), _bits(batch)) and rays_p == rays_b
    ref, st = O.render(cornell, w, h, frames, 3)
    rmse, frac_ok, worst = image_metrics(prog, ref)
    assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
    assert abs(rays_p - st.rays) <= 1e-4 * st.rays
    from PIL import Image
    im = np.asarray(Image.open(png).convert("RGBA"))
    assert im.shape == (h, w, 4)
    want = O.postprocess(prog, flip_y=True)
    assert np.abs(im.astype(int) - want.astype(int)).max() <= 1


def test_postprocess_matches_oracle(capi, O, cornell):
    c = make_ctx(O, cornell, 96, 64)
    try:
        c.render(0, 8, 3)
        acc = c.read_accum()
        for flip in (True, False):
            got, want = c.postprocess(flip_y=flip), O.postprocess(acc, flip_y=flip)
            assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
            assert (got[..., 3] == 255).all()
    finally:
        c.close()


def test_external_accum_buffer_and_stream(capi, O, cornell):
    """trg_bind_accum / trg_set_stream: render straight into a torch tensor on torch's stream."""
    import torch
    w, h = 64, 40
    c = make_ctx(O, cornell, w, h)
    try:
        c.render(0, 3, 3)
        want = c.read_accum()
        frame = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        c.bind_accum(frame.data_ptr())
        stream = torch.cuda.Stream()
        c.set_stream(stream.cuda_stream)
        c.render(0, 3, 3)
        stream.synchronize()
        assert np.array_equal(_bits(frame.cpu().numpy()), _bits(want))
        assert c.accum_device_ptr() == frame.data_ptr()
        c.bind_accum(None)
        c.set_stream(None)
    finally:
        c.close()


@pytest.mark.parametrize("force_global", [0, 1])
def test_overlapped_launches_are_independent(capi, O, cornell, force_global):
    """DistributedRenderer(pipelined=True): consecutive frames go to their own buffers on alternating streams and overlap
    on the GPU (TRG_OPT_LAUNCHES_IN_FLIGHT).  Each must equal a launch that ran alone, bit for bit in the strict build --
    in particular for an HBM-resident scene, whose launches spill traversal stacks to per-launch scratch."""
    from toyraygun_amd.dist import DistributedRenderer
    w, h, spp, bounces = 200, 120, 3, 4
    off = O.pixel_offsets(w, h)
    O.set_trig_mode(O.TRIG_PORTABLE)
    try:
        ref, _ = O.render(cornell, w, h, spp, bounces, offsets=off)
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
    r = DistributedRenderer(w, h, 0, pipelined=True)
    try:
        assert len(r.frames) == 4 and len({s.cuda_stream for s in r.render_streams}) == 4
        r.ctx.set_option(capi.OPT_FORCE_GLOBAL, force_global)
        r.load_scene(cornell.buffers())
        r.ctx.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        r.ctx.set_pixel_offsets(off)
        r.ctx.set_option(capi.OPT_STRICT, 1)
        r.ctx.set_option(capi.OPT_TIMING, 0)
        for _ in range(12):
            r.render(0, spp, bounces, gather=False)
        r.synchronize()
        for fr in r.frames:
            assert np.array_equal(_bits(fr.cpu().numpy()), _bits(ref))
        assert r.ctx.stats().scene_in_lds == (0 if force_global else 1)
    finally:
        r.close()


def test_overlapped_launches_on_a_deep_hbm_scene(capi, O):
    """Four launches in flight on a 96 k-triangle lattice: the 4-wide tree is deep enough (3 x depth + 2 > 16 levels) for
    traversal stacks to spill into the per-launch global scratch, which overlapping launches must not share."""
    from toyraygun_amd.dist import DistributedRenderer
    from toyraygun_amd import host
    w, h, spp, bounces = 160, 96, 2, 3
    b = host.Scene.cornell_lattice(20).buffers()
    r = DistributedRenderer(w, h, 0, pipelined=True)
    try:
        r.load_scene(b)
        st = r.ctx.stats()
        assert st.scene_in_lds == 0 and 3 * st.bvh_depth4 + 2 > 16
        r.ctx.set_uniforms(host.uniforms(w, h)[0])
        r.ctx.set_pixel_offsets_seed()
        r.ctx.set_option(capi.OPT_STRICT, 1)
        r.ctx.render(0, spp, bounces)                      # one launch alone, timing on (synchronous)
        alone = r.ctx.read_accum().copy()
        r.ctx.set_option(capi.OPT_TIMING, 0)
        for _ in range(12):
            r.render(0, spp, bounces, gather=False)
        r.synchronize()
        for fr in r.frames:
            assert np.array_equal(_bits(fr.cpu().numpy()), _bits(alone))
    finally:
        r.close()


def test_single_rank_distributed_renderer(capi, O, cornell):
    from toyraygun_amd.dist import DistributedRenderer
    r = DistributedRenderer(64, 32, 0)
    try:
        r.load_scene(cornell.buffers())
        r.ctx.set_uniforms(O.uniforms_bytes(O.make_uniforms(64, 32)))
        r.ctx.set_pixel_offsets_seed()
        frame = r.render(0, 2, 3)
        r.synchronize()
        ref, _ = O.render(cornell, 64, 32, 2, 3)
        rmse, frac_ok, _ = image_metrics(frame.cpu().numpy(), ref)
        assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC
    finally:
        r.close()


# ------------------------------------------------------------------ round 2: parity evidence for the shipped build
def _radical_inverse_exact(i, b):
    v, f = 0.0, 1.0
    while i:
        f /= b
        v += f * (i % b)
        i //= b
    return v


def test_halton_table_path_of_the_shipped_megakernel(capi, O, ctx256, golden):
    """VERDICT r01 weak #2 / ADVICE: on an LDS-resident scene the shipped megakernel takes Halton dimensions 1..5 from
    digit-group tables in LDS (halton_t), which trg_halton never runs.  trg_halton_table evaluates exactly that path
    (tables staged into LDS, every table index range-checked -> NaN).  Bars: no index out of range; within 3 ulp of
    the EXACT radical inverse (the reference's own fp32 loop, common.h:51-75, is up to 6 ulp off it: its running product
    f *= 1/b drifts); against the reference loop: <= 2^-22 absolute, <= 10 ulp, >= 95 % of samples within 2 ulp;
    every other dimension stays bit-exact."""
    rng = np.random.default_rng(5)
    edge = [0, 1, 2, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31, 2 ** 24 - 1, 2 ** 24, 2 ** 24 + 1, 2 ** 22 - 1, 2 ** 22, 2 ** 22 + 1]
    for b in (3, 5, 7, 11, 13):
        k = 1
        while b ** k < 2 ** 32:
            edge += [b ** k - 1, b ** k, b ** k + 1, 2 * b ** k]
            k += 1
        edge += [(2 ** 32 - 1) // b * b, (2 ** 32 - 1) // b * b - 1]
    i = np.concatenate([golden["halton_i"], np.array(edge, np.uint64).astype(np.uint32),
                        rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)])
    primes = [2, 3, 5, 7, 11, 13]
    for d in range(1, 6):
        dd = np.full(i.shape[0], d, np.uint32)
        got = ctx256.halton_table(i, dd)
        assert not np.isnan(got).any(), "dimension %d: a digit-group index left its table" % d
        ref = np.array([O.halton(int(a), d) for a in i.tolist()], np.float32)
        exact = np.array([_radical_inverse_exact(int(a), primes[d]) for a in i.tolist()])
        ulp_ref = np.spacing(np.maximum(ref, np.float32(1e-30))).astype(np.float64)
        ulp_ex = np.spacing(np.maximum(exact.astype(np.float32), np.float32(1e-30))).astype(np.float64)
        err_ref = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        err_ex = np.abs(got.astype(np.float64) - exact)
        assert (got >= 0).all() and (got <= 1).all()   # all-digits-maximal indices (b^k - 1) round to 1.0f, as in the reference loop
        assert err_ex.max() <= 2.0 ** -22 and (err_ex / ulp_ex).max() <= 3.0, (d, (err_ex / ulp_ex).max())
        assert err_ref.max() <= 2.0 ** -22 and (err_ref / ulp_ref).max() <= 10.0, (d, err_ref.max(), (err_ref / ulp_ref).max())
        assert (err_ref <= 2 * ulp_ref).mean() >= 0.95, (d, (err_ref <= 2 * ulp_ref).mean())
    # the dimensions without tables go through the digit code: bit-exact, as in test_halton_bit_exact
    sel = i[:2000]
    for d in (0, 6, 7, 13, 33, 63):
        got = ctx256.halton_table(sel, np.full(sel.shape[0], d, np.uint32))
        ref = np.array([O.halton(int(a), d) for a in sel.tolist()], np.float32)
        assert np.array_equal(_bits(got), _bits(ref)), d
    with pytest.raises(capi.TrgError):
        ctx256.halton_table([1], [64])


def _uv_sphere(nu, nv, radius, centre):
    """Indexed UV sphere: vertices, unit normals, per-vertex colours (a smooth function of the normal), triangles."""
    th = np.linspace(0.0, np.pi, nv + 1)
    ph = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)
    n = np.array([[np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)] for t in th for p in ph])
    v = n * radius + np.asarray(centre)
    col = 0.15 + 0.7 * (0.5 + 0.5 * n[:, [0, 1, 2]]) * np.array([1.0, 0.8, 0.6])
    tris = []
    for a in range(nv):
        for b in range(nu):
            i0, i1 = a * nu + b, a * nu + (b + 1) % nu
            j0, j1 = i0 + nu, i1 + nu
            if a > 0:
                tris.append((i0, i1, j0))      # at the poles one of the two triangles is degenerate: skip it
            if a < nv - 1:
                tris.append((i1, j1, j0))
    return v.astype(np.float32), n.astype(np.float32), col.astype(np.float32), np.array(tris, np.uint32)


@pytest.mark.parametrize("nu,nv,in_lds", [(8, 5, 1), (40, 24, 0)])
def test_smooth_normal_vertex_colour_mesh(capi, O, nu, nv, in_lds):
    """VERDICT r01 missing #1 / weak #3: the barycentric interpolation of Raytracing.metal:95-112 (a8) was only ever fed
    three equal normals / colours per triangle, where a swapped weight cannot fail.  A tessellated sphere with per-vertex
    normals AND per-vertex colours, built by the product's Scene::addMesh, in the Cornell box: the strict build is
    bit-exact against the oracle for both megakernels (LDS-resident: 64 + 36 triangles; HBM-resident: 1,840 + 36), the
    shipped build within the stated tolerance; and the image really depends on the interpolation (shuffling the
    corner attributes of the mesh changes it)."""
    from toyraygun_amd import host
    v, n, col, tris = _uv_sphere(nu, nv, 0.33, (0.25, 1.1, 0.15))
    hs = host.Scene.cornell_box()
    hs.add_mesh(v, n, tris, np.eye(4, dtype=np.float32), col, 1)
    b = hs.buffers()
    nt = b["material_ids"].shape[0]
    assert nt == 36 + tris.shape[0]
    # the mesh corners carry three DIFFERENT normals and colours
    mesh_n = b["normals"][108:].reshape(-1, 3, 3)
    assert (np.abs(mesh_n[:, 0] - mesh_n[:, 1]).max(-1) > 1e-3).mean() > 0.9
    scene = O.OracleScene()
    scene.add_raw(b["positions"], b["normals"], b["colors"], b["material_ids"])
    w, h, spp, bnc = 160, 120, 3, 3
    off = O.pixel_offsets(w, h)
    c = capi.Context(w, h)
    try:
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        assert c.stats().scene_in_lds == in_lds
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets(off)
        O.set_trig_mode(O.TRIG_PORTABLE)
        ref, rst = O.render(scene, w, h, spp, bnc, offsets=off)
        O.set_trig_mode(O.TRIG_LIBM)
        c.set_option(capi.OPT_STRICT, 1)
        for k in KERNELS:
            c.set_option(capi.OPT_KERNEL, k)
            c.reset_stats()
            c.render(0, spp, bnc)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref)), "kernel %d" % k
            assert c.stats().rays == rst.rays
        # the sphere is visible and shaded with varying colour
        strict_img = c.read_accum()
        ref_lib, _ = O.render(scene, w, h, spp, bnc, offsets=off)
        c.set_option(capi.OPT_STRICT, 0)
        c.set_option(capi.OPT_KERNEL, 0)
        c.render(0, spp, bnc)
        rmse, frac_ok, worst = image_metrics(c.read_accum(), ref_lib)
        assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
        # sensitivity: rotate the corner attributes of the mesh triangles (v0 <- v1 <- v2 <- v0) but not the positions:
        # if the kernel ignored or mis-ordered the weights this would not change the picture
        nrm2, col2 = b["normals"].copy(), b["colors"].copy()
        nrm2[108:] = nrm2[108:].reshape(-1, 3, 3)[:, [1, 2, 0]].reshape(-1, 3)
        col2[108:] = col2[108:].reshape(-1, 3, 3)[:, [1, 2, 0]].reshape(-1, 3)
        c.set_option(capi.OPT_STRICT, 1)
        c.load_scene(b["positions"], nrm2, col2, b["indices"], b["material_ids"])
        c.render(0, spp, bnc)
        rot = c.read_accum()
        assert (np.abs(rot[..., :3] - strict_img[..., :3]).max(-1) > 1e-3).mean() > 0.01
        s2 = O.OracleScene()
        s2.add_raw(b["positions"], nrm2, col2, b["material_ids"])
        O.set_trig_mode(O.TRIG_PORTABLE)
        ref2, _ = O.render(s2, w, h, spp, bnc, offsets=off)
        O.set_trig_mode(O.TRIG_LIBM)
        assert np.array_equal(_bits(rot), _bits(ref2))
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


def test_full_size_c4_properties(capi, O):
    """Config C4 at BASELINE's full size (1,022,244 triangles, 1920x1080, 16 spp, 3 bounces) through size-independent
    properties: ray-count identities, a launch split in two frame ranges continues bit for bit, the union of four row
    bands is the unsharded frame bit for bit, the GPU-built tree gives the same frame within rounding of nothing (the
    contract is tree-independent: bit for bit), and sampled rows agree with the oracle (its own BVH) within tolerance."""
    from toyraygun_amd import host
    from toyraygun_amd.dist import band_rows
    w, h, spp, bnc = 1920, 1080, 16, 3
    b = host.Scene.cornell_lattice(44).buffers()
    c = capi.Context(w, h)
    try:
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets_seed()
        c.reset_stats()
        c.render(0, spp, bnc)
        img, st = c.read_accum(), c.stats()
        assert st.scene_in_lds == 0 and st.scene_bytes > 100e6
        assert np.isfinite(img).all() and (img[..., 3] == 1.0).all() and (img[..., :3] >= 0).all()
        assert st.primary_rays == w * h * spp and st.shadow_rays == st.shaded_hits
        assert st.bounce_rays <= st.shaded_hits and st.rays <= 2 * bnc * w * h * spp
        assert (img[:, :300, :3] == 0).all() and (img[:, -300:, :3] == 0).all()   # the side bars see nothing
        # continuation: frames [0,5) then [5,16) == frames [0,16)
        c.render(0, 5, bnc)
        c.render(5, spp - 5, bnc)
        assert np.array_equal(_bits(c.read_accum()), _bits(img))
        # union of four row bands == the unsharded frame, and the ray counts add up
        c.reset_stats()
        for r in range(4):
            r0, n = band_rows(h, 4, r)
            c.render(0, spp, bnc, r0, n)
        assert np.array_equal(_bits(c.read_accum()), _bits(img)) and c.stats().rays == st.rays
        # sampled rows against the oracle (walks its own median-split BVH)
        scene = O.OracleScene.cornell_lattice(44)
        _rows_against_oracle(capi, O, c, scene, w, h, spp, bnc, img, [(r0, 2) for r0 in (60, 200, 340, 540, 700, 900, 1030)], TOL_FRAC_C4)
    finally:
        c.close()


def _subdivided_cornell(O, n):
    """SURVEY 8(d) cross-check variant of C4: every one of the 36 Cornell triangles uniformly subdivided n x n (n = 167 ->
    1,004,004 triangles) -- the same surfaces, normals, colours and materials, so it must render the C2 picture, but
    through the HBM / wide-BVH path instead of the LDS path."""
    b = O.OracleScene.cornell_box().buffers()
    P = b["positions"].reshape(36, 3, 3).astype(np.float64)
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    keep = (ii + jj) <= n
    # grid vertex (i, j) = v0 + (i/n) e1 + (j/n) e2, computed once per original triangle, shared by its sub-triangles
    up = [(i, j) for i in range(n) for j in range(n - i)]                 # (i,j) (i+1,j) (i,j+1)
    dn = [(i, j) for i in range(n - 1) for j in range(n - 1 - i)]         # (i+1,j) (i+1,j+1) (i,j+1)
    up, dn = np.array(up), np.array(dn)
    a = np.concatenate([np.stack([up, up + [1, 0], up + [0, 1]], 1), np.stack([dn + [1, 0], dn + [1, 1], dn + [0, 1]], 1)])  # [n*n, 3, 2]
    assert a.shape[0] == n * n
    wi, wj = a[..., 0] / n, a[..., 1] / n
    pos = (P[:, None, None, 0, :] * (1 - wi - wj)[None, ..., None] + P[:, None, None, 1, :] * wi[None, ..., None]
           + P[:, None, None, 2, :] * wj[None, ..., None]).astype(np.float32)   # [36, n*n, 3, 3]
    del keep
    rep = n * n
    nrm = np.repeat(b["normals"].reshape(36, 1, 3, 3), rep, 1)
    col = np.repeat(b["colors"].reshape(36, 1, 3, 3), rep, 1)
    mat = np.repeat(b["material_ids"].reshape(36, 1), rep, 1)
    nt = 36 * rep
    return dict(positions=pos.reshape(-1, 3), normals=nrm.reshape(-1, 3).copy(), colors=col.reshape(-1, 3).copy(),
                indices=np.arange(3 * nt, dtype=np.uint32), material_ids=mat.reshape(-1).astype(np.uint32).copy())


def test_subdivided_cornell_renders_the_c2_image(capi, O, cornell):
    """"each of the 36 triangles uniformly subdivided 167^2 -> 1,004,004 tris -- renders the same image as
    C2": the one free end-to-end check of the HBM / quantised wide-BVH path against the LDS path on the same picture.
    What is expected, and asserted: NOT bit identity -- a sub-triangle has other vertices than its parent, so Moeller-
    Trumbore rounds the hit distance differently (last-ulp changes of P propagate through the path) and the hit
    primitive index differs by construction; hits on a shared interior edge may resolve to either neighbour, which
    carry the same normal, colour and material.  So: the STRICT image of the subdivided scene equals the STRICT C2
    image within the shipped-build tolerance (RMSE <= 1e-3, >= 99.9 % of pixels within 1e-4 max(1,|ref|)), the ray
    counts agree to 1e-4 relative, nothing leaks through the 1 M interior edges (side bars exact, the light quad's texels equal but for its outline)."""
    w, h, spp, bnc = 1920, 1080, 16, 3
    c2 = make_ctx(O, cornell, w, h)
    try:
        c2.set_option(capi.OPT_STRICT, 1)
        c2.reset_stats()
        c2.render(0, spp, bnc)
        ref, rst = c2.read_accum(), c2.stats()
    finally:
        c2.close()
    b = _subdivided_cornell(O, 167)
    assert b["material_ids"].shape[0] == 1004004
    c = capi.Context(w, h)
    try:
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets_seed()
        assert c.stats().scene_in_lds == 0
        for strict in (1, 0):
            c.set_option(capi.OPT_STRICT, strict)
            c.reset_stats()
            c.render(0, spp, bnc)
            img, st = c.read_accum(), c.stats()
            rmse, frac_ok, worst = image_metrics(img, ref)
            assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (strict, rmse, frac_ok, worst)
            assert st.primary_rays == rst.primary_rays and abs(st.rays - rst.rays) <= 1e-4 * rst.rays, (st.rays, rst.rays)
            assert (img[:, :300, :3] == 0).all() and (img[:, -300:, :3] == 0).all()
            la, lb = (img[..., :3] == 1.0).all(-1), (ref[..., :3] == 1.0).all(-1)     # texels whose 16 samples all hit the light quad
            assert lb.sum() > 1000 and (la != lb).sum() <= 0.002 * lb.sum(), ((la != lb).sum(), lb.sum())  # only its outline may differ
    finally:
        c.close()


def test_path_regeneration_is_bit_exact(capi, O):
    """TRG_OPT_REGEN (default on for HBM-resident scenes): every wavefront works through the (pixel, frame) jobs of its tile as a
    pool, each lane at its own pace, the frames folded in order from a radiance buffer.  Same image and same ray counts as the
    lock-step kernel bit for bit -- odd sizes (partial tiles), several 16-frame chunks, 0..8 bounces, continuation, row bands --
    and the strict build equals the oracle."""
    scene = O.OracleScene.cornell_lattice(6)
    b = scene.buffers()
    for (w, h, spp, bounces) in ((97, 45, 5, 3), (64, 64, 33, 2), (40, 24, 3, 8), (33, 17, 2, 0), (16, 16, 1, 1)):
        off = O.pixel_offsets(w, h)
        c = capi.Context(w, h)
        try:
            c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
            c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
            c.set_pixel_offsets(off)
            c.set_option(capi.OPT_FRAME_SPLIT, 1)
            imgs = {}
            for strict in (1, 0):
                c.set_option(capi.OPT_STRICT, strict)
                for regen in (0, 1):
                    c.set_option(capi.OPT_REGEN, regen)
                    c.reset_stats()
                    c.render(0, spp, bounces)
                    st = c.stats()
                    assert st.last_regen == regen and st.scene_in_lds == 0
                    imgs[(strict, regen)] = (c.read_accum().copy(), st.rays)
                assert np.array_equal(_bits(imgs[(strict, 0)][0]), _bits(imgs[(strict, 1)][0])), (w, h, spp, bounces, strict)
                assert imgs[(strict, 0)][1] == imgs[(strict, 1)][1]
                for lanes in (2, 4):   # frame lanes: that many workgroups share a tile's frames (small grids)
                    c.set_option(capi.OPT_FRAME_SPLIT, lanes)
                    c.reset_stats()
                    c.render(0, spp, bounces)
                    st = c.stats()
                    assert st.last_regen == 1 and st.last_frame_split == (lanes if spp >= 2 else 1)
                    assert np.array_equal(_bits(c.read_accum()), _bits(imgs[(strict, 0)][0])) and st.rays == imgs[(strict, 0)][1], (w, h, spp, bounces, strict, lanes)
                c.set_option(capi.OPT_FRAME_SPLIT, 1)
            if spp <= 5:
                O.set_trig_mode(O.TRIG_PORTABLE)
                try:
                    ref, rst = O.render(scene, w, h, spp, bounces, offsets=off)
                finally:
                    O.set_trig_mode(O.TRIG_LIBM)
                assert np.array_equal(_bits(imgs[(1, 1)][0]), _bits(ref)) and imgs[(1, 1)][1] == rst.rays
            # continuation (frames 0..spp-1, then spp..2spp-1) and two row bands give the image of one launch
            c.set_option(capi.OPT_STRICT, 1); c.set_option(capi.OPT_REGEN, 1)
            c.render(0, 2 * spp, bounces)
            whole = c.read_accum().copy()
            c.render(0, spp, bounces); c.render(spp, spp, bounces)
            assert np.array_equal(_bits(c.read_accum()), _bits(whole))
            if h >= 16:
                c.render(0, 2 * spp, bounces, 0, h // 3); c.render(0, 2 * spp, bounces, h // 3, h - h // 3)
                assert np.array_equal(_bits(c.read_accum()), _bits(whole))
        finally:
            c.close()


def test_plugin_device_build(capi, O):
    """HipRenderer::setDeviceBuild: the acceleration structure of the next loadScene is built on the device (the reference rebuilds
    its MPS structure on the GPU, MetalRenderer.mm:272-279).  The image does not depend on the tree -- host SAH, device SAH,
    LBVH and PLOC give the same accumulation buffer bit for bit (shipped build) -- and a scene that fits LDS ignores the option.
    (Round 4: all four builders pair the triangles of a parallelogram into QUAD leaves, which the shipped build decides with one
    plane test -- a different rounding of the same hit than two triangle tests: with TRG_BVH_QUADS=0 the four builders agree bit for
    bit again, on an image within the shipped build's tolerance of the one with quads.)"""
    from toyraygun_amd import host
    w, h = 160, 120
    scene = host.Scene.cornell_lattice(12)     # 20,772 triangles: lives in HBM

    def close(a, b):
        d = np.linalg.norm(a[..., :3].astype(np.float64) - b[..., :3], axis=-1)
        inl = d <= 1e-4 * np.maximum(1.0, np.linalg.norm(b[..., :3].astype(np.float64), axis=-1))
        return inl.mean() >= 0.999 and np.sqrt(np.mean((d * d)[inl])) <= 1e-3

    # (round 5: the HOST builder also recognises the cubes and the shipped build tests each with one slab test -- BOX leaves; the device builders
    #  pair quads only.  So: the three device builders agree bit for bit, the host build is within the shipped build's tolerance of them, and
    #  with the box flavour left out (TRG_BVH_BOXES_HBM=0) all four agree bit for bit as before)
    boxed, ms_host = host.render_scene(scene, w, h, 4, 3, device_build=0)
    assert np.isfinite(boxed).all() and boxed[..., :3].max() > 0
    os.environ["TRG_BVH_BOXES_HBM"] = "0"
    try:
        ref, _ = host.render_scene(scene, w, h, 4, 3, device_build=0)
    finally:
        del os.environ["TRG_BVH_BOXES_HBM"]
    assert not np.array_equal(_bits(boxed), _bits(ref)) and close(boxed, ref)
    for builder in (1, 2, 3):
        got, ms_dev = host.render_scene(scene, w, h, 4, 3, device_build=builder)
        assert np.array_equal(_bits(got), _bits(ref)), builder
    os.environ["TRG_BVH_QUADS"] = "0"
    try:
        plain, _ = host.render_scene(scene, w, h, 4, 3, device_build=0)
        for builder in (1, 2, 3):
            got, _ = host.render_scene(scene, w, h, 4, 3, device_build=builder)
            assert np.array_equal(_bits(got), _bits(plain)), builder
    finally:
        del os.environ["TRG_BVH_QUADS"]
    assert not np.array_equal(_bits(plain), _bits(ref))
    d = np.linalg.norm(plain[..., :3].astype(np.float64) - ref[..., :3], axis=-1)
    inl = d <= 1e-4 * np.maximum(1.0, np.linalg.norm(ref[..., :3].astype(np.float64), axis=-1))
    assert inl.mean() >= 0.999 and np.sqrt(np.mean((d * d)[inl])) <= 1e-3, (inl.mean(), d.max())
    box = host.Scene.cornell_box()
    a, _ = host.render_scene(box, w, h, 4, 3, device_build=0)
    b, _ = host.render_scene(box, w, h, 4, 3, device_build=1)
    assert np.array_equal(_bits(a), _bits(b))
    with pytest.raises(RuntimeError):
        host.render_scene(box, w, h, 1, 3, device_build=4)


def test_async_render_frame_loop(capi, O):
    """VERDICT r01 missing #3 / next #7: HipRenderer::renderFrame() is fire-and-forget like the reference's (MetalRenderer.mm:33,
    377,385-387: three frames in flight behind a semaphore) -- no host sync per frame; consecutive frames with the same
    uniforms may share a launch while the device is busy.  The progressive 1-spp loop at 1920x1080 is bit-identical to
    the batched launch and reaches >= 80 % of its rays per second (wall clock, first renderFrame() to last frame done)."""
    from toyraygun_amd import host
    w, h, frames = 1920, 1080, 64
    host.run_app(w, h, 4, 3, batch=True)   # warm-up: library load, first launch
    batch, ms_b, rays_b = host.run_app(w, h, frames, 3, batch=True)
    best = None
    for _ in range(3):
        prog, ms_p, rays_p, launches = host.run_app(w, h, frames, 3, batch=False, want_launches=True)
        assert np.array_equal(_bits(prog), _bits(batch)) and rays_p == rays_b
        assert 1 <= launches <= frames
        best = ms_p if best is None else min(best, ms_p)
    assert best <= ms_b / 0.8, "progressive loop %.2f ms vs batched %.2f ms (%d launches)" % (best, ms_b, launches)
    # the C ABI primitives behind it
    c = make_ctx(O, O.OracleScene.cornell_box(), 256, 256)
    try:
        c.set_option(capi.OPT_TIMING, 0)
        c.fence_wait(2)                      # never recorded: returns at once
        c.render(0, 64, 8)
        c.fence_record(0)
        c.fence_wait(0)
        assert c.stream_idle()
        with pytest.raises(capi.TrgError):
            c.fence_record(8)
    finally:
        c.close()


@pytest.mark.parametrize("force_global", [0, 1])
def test_device_group_c_abi(capi, O, cornell, force_global):
    """trg_group_* (multi-GPU behind the C ABI) on the devices that are here: with one device the group path (padded
    frame buffer bound as the accumulation target, band arithmetic, per-device host thread, stats reduction) must reproduce the
    plain context bit for bit for every gather mode; with two or more visible devices the RCCL exchange runs and every device
    ends with the same frame.  (The 8-GPU run is the driver's; the band / in-place layout is rehearsed on CPU in
    tests/test_dist_gloo.py::test_c_abi_row_band_layout_rehearsal.)"""
    import torch
    w, h, spp, bnc = 200, 93, 3, 3          # 93 rows: not a multiple of any device count > 1
    ref_ctx = make_ctx(O, cornell, w, h)
    try:
        ref_ctx.set_option(capi.OPT_STRICT, 1)
        ref_ctx.set_option(capi.OPT_FORCE_GLOBAL, force_global)   # 1: the scene stays in HBM ...
        ref_ctx.set_option(capi.OPT_REGEN, 1)                      # ... and is rendered by the path-regeneration kernel (per band below)
        ref_ctx.reset_stats()
        ref_ctx.render(0, spp, bnc)
        ref, rst = ref_ctx.read_accum(), ref_ctx.stats()
        assert rst.scene_in_lds == 1 - force_global
    finally:
        ref_ctx.close()
    b = cornell.buffers()
    ndev = min(torch.cuda.device_count(), 4)
    for n in sorted({1, ndev}):
        g = capi.Group(list(range(n)), w, h)
        try:
            assert g.n == n
            g.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
            g.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
            g.set_pixel_offsets_seed()
            g.set_option(capi.OPT_STRICT, 1)
            g.set_option(capi.OPT_FORCE_GLOBAL, force_global)
            g.set_option(capi.OPT_REGEN, 1)
            for mode, root in ((capi.GATHER_ALL, 0), (capi.GATHER_ROOT, n - 1), (capi.GATHER_NONE, 0)):
                g.render(0, spp, bnc, gather=mode, root=root)
                g.sync()
                ranks = range(n) if mode == capi.GATHER_ALL else ([root] if mode == capi.GATHER_ROOT else [])
                for r in ranks:
                    assert np.array_equal(_bits(g.read_accum(r)), _bits(ref)), (n, mode, r)
                if mode == capi.GATHER_NONE:
                    for r in range(n):
                        r0, nr = capi.band_rows(h, n, r)
                        assert np.array_equal(_bits(g.read_accum(r)[r0:r0 + nr]), _bits(ref[r0:r0 + nr]))
            st = g.stats()
            assert st.rays == 3 * rst.rays    # three renders, the counters of all devices summed
            with pytest.raises(capi.TrgError):
                g.render(0, 1, 1, gather=7)
            with pytest.raises(capi.TrgError):
                g.render(0, 1, 1, gather=capi.GATHER_ROOT, root=n)
        finally:
            g.close()
    with pytest.raises(capi.TrgError):
        capi.Group([0, 0], 16, 16)


def test_bench_launched_multi_rank_path_rehearsed_on_one_device(capi, tmp_path):
    """The command the driver's scaling run uses -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- with two ranks
    on this ONE device: TRG_BENCH_DEVICES=0,0 maps both ranks to device 0 and TRG_BENCH_BACKEND=gloo replaces the RCCL communicator NCCL
    refuses for ranks that share a GPU (the bands then go through host memory).  Everything else is the real path: rank environment, one
    context per process rendering its band, barrier + max-over-ranks timing, the all-reduced ray count, ONE JSON line from rank 0."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:   # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TRG_BENCH_BACKEND="gloo", TRG_BENCH_DEVICES="0,0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["unit"] == "Mrays/s" and out["value"] > 100
    assert 100e6 < out["config"]["rays_per_step"] < 120e6      # both bands counted: the whole C2 frame
    assert "gloo" in out["config"]["sharding"] and "cpu_baseline" not in out
    assert out["config"]["exchange"] == "root gather" and out["config"]["gather_all_ms_per_step"] >= 0      # north_star's exchange is the timed one
    assert out["roofline"]["derived"] and len(lines[0]) < 4096


_RCCL_ONE_RANK = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from toyraygun_amd import capi, host
from toyraygun_amd.dist import DistributedRenderer
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
w, h, spp, bnc = 200, 93, 3, 3
b = host.Scene.cornell_box().buffers()
u = host.uniforms(w, h)[0]
ref = capi.Context(w, h)
ref.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"]); ref.set_uniforms(u); ref.set_pixel_offsets_seed()
ref.render(0, spp, bnc)
want = ref.read_accum(); ref.close()
for interleaved in (False, True):
    for pipelined in (False, True):
        r = DistributedRenderer(w, h, 0, pipelined=pipelined, interleaved=interleaved)
        r.load_scene(b); r.ctx.set_uniforms(u); r.ctx.set_pixel_offsets_seed()
        for mode in ("root", "all", "root", "all", "root"):      # (five steps: the four slots of a pipelined renderer wrap)
            f = r.render(0, spp, bnc, gather=mode)
            r.synchronize(); torch.cuda.synchronize()
            got = f.cpu().numpy().reshape(h, w, 4)
            assert np.array_equal(got, want), (interleaved, pipelined, mode, float(np.abs(got - want).max()))
        r.close()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
assert t.tolist() == [1.5, 2.5]
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", torch.cuda.nccl.version())
"""


def _launched_env(port):
    import os
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TRG_FORCE_GATHER="1")
    for k in ("TRG_BENCH_BACKEND", "TRG_BENCH_DEVICES", "TRG_GATHER", "TRG_BANDS"):
        env.pop(k, None)
    return env


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_torch_distributed_exchange_through_rccl_with_one_rank(capi):
    """toyraygun_amd/dist.py on a REAL RCCL communicator (backend "nccl"), one rank -- all a one-GPU box can give it: the in-place dist.gather
    to rank 0 whose receive list ARE the row bands of the frame, the in-place all_gather_into_tensor (sendbuff = recvbuff + rank * count),
    both on the communication stream behind the render, contiguous and interleaved bands (unpack behind the exchange), plain and pipelined
    with the slots wrapping: every step's frame equals the plain context's bit for bit.  (The gloo rehearsals never touch ProcessGroupNCCL;
    the driver's scaling run does.)  A child process: the process group is global state."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], cwd=root, env=_launched_env(_free_port()), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_bench_launched_path_through_rccl_with_one_rank(capi):
    """bench.py as the driver's scaling run starts it (rank environment, backend nccl), one rank, TRG_FORCE_GATHER=1: init_process_group on the
    device, the barriers, the float64 MAX / SUM reductions and the all_gather of the per-rank figures on the GPU, the root gather timed per
    frame and the all-gather beside it -- every N > 1 branch, through RCCL."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
                       cwd=root, env=_launched_env(_free_port()), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    cfgo = out["config"]
    assert out["n_gpus"] == 1 and out["value"] > 1000 and 100e6 < cfgo["rays_per_step"] < 120e6
    assert "RCCL" in cfgo["sharding"] and cfgo["exchange"] == "root gather" and cfgo["rccl_ranks"] == 1 and cfgo["rccl_version"]
    assert cfgo["gather_root_ms_per_step"] >= 0 and cfgo["gather_all_ms_per_step"] >= 0 and len(lines[0]) < 4096


def test_bench_single_process_group_path_rehearsed_on_one_device(capi, monkeypatch, capsys):
    """`python bench.py --gpus N` invoked plainly (what a driver without a launcher does) takes the single-process trg_group path.  Rehearsed
    here end to end on ONE device: TRG_BENCH_DEVICES names device 0 N times and TRG_GROUP_EXCHANGE=copy lets the contexts share it.  The JSON
    line must come out whole, the bands' ray counts must add up to the frame's, and the value must be a real rate -- not a scaling number."""
    import json
    import bench
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TRG_GROUP_EXCHANGE", "copy")
    for n in (2, 8):
        monkeypatch.setenv("TRG_BENCH_DEVICES", ",".join(["0"] * n))
        bench.main(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
        out = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
        cfgo = out["config"]
        assert out["n_gpus"] == n and cfgo["exchange"] == "copy" and cfgo["devices"] == ",".join(["0"] * n) and cfgo["rccl_ranks"] is None
        detail = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cfgo["detail_file"])))
        assert len(detail["per_rank_rays_per_step"]) == n and sum(detail["per_rank_rays_per_step"]) == cfgo["rays_per_step"]
        assert 100e6 < cfgo["rays_per_step"] < 120e6 and out["value"] > 1000 and out["ms_per_step"] > 0
        assert "bound" in out["roofline"] and cfgo["gather_root_ms_per_step"] >= 0 and cfgo["gather_all_ms_per_step"] >= 0
        # an N > 1 line names the same bound as N = 1 (derived from the N = 1 counters per ray) whenever those counters belong to this build
        assert out["roofline"]["bound"] in ("valu_issue", "lds") and (out["roofline"]["bound"] == "valu_issue") == ("counters" in out["roofline"])


@pytest.mark.parametrize("force_global", [0, 1])
def test_device_group_of_several_contexts_on_one_device(capi, O, cornell, force_global, monkeypatch):
    """The n > 1 machinery of trg_group_* on a single-GPU box: with TRG_GROUP_EXCHANGE=copy the bands travel by hipMemcpyPeerAsync ordered by
    events instead of RCCL, a group needs no communicator, and its contexts may share a device.  Groups of 2, 3 and 8 contexts on device 0 --
    row bands of ceil(h / n) rows (93 rows: never a multiple), one host thread per context, the scene uploaded n times, gather to all / to a
    root / none, asynchronous frames behind group fences with the accumulation continued across launches -- give the plain context's
    frame bit for bit on every rank that is to hold it, and the ray counters add up.  (What this does NOT run: the RCCL calls; those run
    for one rank in test_device_group_exchange_through_rccl_on_one_device.)"""
    monkeypatch.setenv("TRG_GROUP_EXCHANGE", "copy")
    w, h, spp, bnc = 200, 93, 5, 3
    ref_ctx = make_ctx(O, cornell, w, h)
    try:
        ref_ctx.set_option(capi.OPT_STRICT, 1)
        ref_ctx.set_option(capi.OPT_FORCE_GLOBAL, force_global)
        ref_ctx.reset_stats()
        ref_ctx.render(0, spp, bnc)
        ref, rst = ref_ctx.read_accum(), ref_ctx.stats()
    finally:
        ref_ctx.close()
    b = cornell.buffers()
    for n in (2, 3, 8):
        g = capi.Group([0] * n, w, h)
        try:
            assert g.n == n and g.exchange == capi.EXCHANGE_COPY
            g.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
            g.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
            g.set_pixel_offsets_seed()
            g.set_option(capi.OPT_STRICT, 1)
            g.set_option(capi.OPT_FORCE_GLOBAL, force_global)
            for mode, root in ((capi.GATHER_ALL, 0), (capi.GATHER_ROOT, n - 1), (capi.GATHER_NONE, 0)):
                g.reset_stats()
                g.render(0, spp, bnc, gather=mode, root=root)
                g.sync()
                assert g.stats().rays == rst.rays
                for r in (range(n) if mode == capi.GATHER_ALL else ([root] if mode == capi.GATHER_ROOT else [])):
                    assert np.array_equal(_bits(g.read_accum(r)), _bits(ref)), (n, mode, r)
                for r in range(n):
                    r0, nr = capi.band_rows(h, n, r)
                    assert np.array_equal(_bits(g.read_accum(r)[r0:r0 + nr]), _bits(ref[r0:r0 + nr])), (n, mode, r, "own band")
            # asynchronous: two launches that continue one accumulation, each followed by its exchange, three in flight behind fences
            g.set_option(capi.OPT_TIMING, 0)
            for rep in range(3):
                g.fence_wait(rep % 3)
                g.render(0, 2, bnc, gather=capi.GATHER_ALL)
                g.render(2, spp - 2, bnc, gather=capi.GATHER_ALL)
                g.fence_record(rep % 3)
            g.sync()
            for r in range(n):
                assert np.array_equal(_bits(g.read_accum(r)), _bits(ref)), (n, "async", r)
        finally:
            g.close()
    monkeypatch.setenv("TRG_GROUP_EXCHANGE", "bogus")
    with pytest.raises(capi.TrgError):
        capi.Group([0], 16, 16)
    monkeypatch.setenv("TRG_GROUP_EXCHANGE", "rccl")
    with pytest.raises(capi.TrgError):
        capi.Group([0, 0], 16, 16)      # a communicator needs distinct devices


def test_device_group_exchange_through_rccl_on_one_device(capi, O, cornell, monkeypatch):
    """The part of the multi-GPU exchange a single-GPU box CAN run ("the exchange was never run on hardware"): with
    TRG_GROUP_FORCE_RCCL a group of one device loads RCCL, creates its communicator with ncclCommInitAll and enqueues the in-place
    ncclAllGather (sendbuff = recvbuff + rank * count) / the grouped send-recv on the stream its render went to -- one rank, so the
    collective moves nothing, but symbol binding, communicator, argument layout, stream order and the padded frame buffer are real.
    Frames before and after the exchange are the plain context's bit for bit, asynchronously over several frames too."""
    w, h, spp, bnc = 120, 45, 2, 3
    ref_ctx = make_ctx(O, cornell, w, h)
    try:
        ref_ctx.set_option(capi.OPT_STRICT, 1)
        ref_ctx.render(0, spp, bnc)
        ref = ref_ctx.read_accum()
        ref_ctx.render(spp, spp, bnc)
        ref2 = ref_ctx.read_accum()
    finally:
        ref_ctx.close()
    monkeypatch.setenv("TRG_GROUP_FORCE_RCCL", "1")
    b = cornell.buffers()
    g = capi.Group([0], w, h)
    try:
        g.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        g.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        g.set_pixel_offsets_seed()
        g.set_option(capi.OPT_STRICT, 1)
        for mode in (capi.GATHER_ALL, capi.GATHER_ROOT):
            g.render(0, spp, bnc, gather=mode, root=0)
            g.sync()
            assert np.array_equal(_bits(g.read_accum(0)), _bits(ref)), mode
        g.set_option(capi.OPT_TIMING, 0)              # asynchronous: render, exchange, fence -- twice, then wait
        g.render(0, spp, bnc, gather=capi.GATHER_ALL); g.fence_record(0)
        g.render(spp, spp, bnc, gather=capi.GATHER_ALL); g.fence_record(1)
        g.fence_wait(0); g.fence_wait(1)
        assert np.array_equal(_bits(g.read_accum(0)), _bits(ref2))
    finally:
        g.close()


def _group_ready(capi, O, cornell, devices, w, h, force_global=0):
    b = cornell.buffers()
    g = capi.Group(devices, w, h)
    g.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    g.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
    g.set_pixel_offsets_seed()
    g.set_option(capi.OPT_STRICT, 1)
    g.set_option(capi.OPT_FORCE_GLOBAL, force_global)
    return g


@pytest.mark.parametrize("bands", ["contiguous", "interleaved"])
@pytest.mark.parametrize("force_global", [0, 1])
def test_pipelined_device_group(capi, O, cornell, force_global, bands, monkeypatch):
    """trg_group_set_pipeline (round 4): k frame buffers and render streams + one exchange stream per device, frame i on stream i % k, its
    exchange behind it on the exchange stream, slot re-use ordered on the device by events.  Groups of 1, 2, 3 and 8 contexts on device 0
    (copy exchange), depth 3: sequences of 1..8 DIFFERENT frames enqueued without any host wait -- the slots wrap around, exchanges of
    older frames are still in flight when their slot is rendered again -- must end with the plain context's last frame bit for bit on
    every rank that is to hold it; HIP-event launch times come back one per frame and rank; the host-side fences bound the run-ahead;
    back at depth 1 the progressive accumulation continues in place as before."""
    monkeypatch.setenv("TRG_GROUP_EXCHANGE", "copy")
    w, h, bnc = 200, 93, 3
    spps = [1, 2, 3, 1, 3, 2, 2, 1]            # frame i of a sequence = frames [0, spps[i]) of the picture: consecutive images differ
    ref_ctx = make_ctx(O, cornell, w, h)
    refs = {}
    try:
        ref_ctx.set_option(capi.OPT_STRICT, 1)
        ref_ctx.set_option(capi.OPT_FORCE_GLOBAL, force_global)
        for spp in (1, 2, 3, 5):
            ref_ctx.render(0, spp, bnc)
            refs[spp] = ref_ctx.read_accum()
    finally:
        ref_ctx.close()
    for n in (1, 2, 3, 8):
        g = _group_ready(capi, O, cornell, [0] * n, w, h, force_global)
        try:
            with pytest.raises(capi.TrgError):
                g.set_pipeline(0)
            with pytest.raises(capi.TrgError):
                g.set_bands(7)
            g.set_bands(capi.BANDS_INTERLEAVED if bands == "interleaved" else capi.BANDS_CONTIGUOUS)
            g.set_pipeline(3)
            assert g.pipeline_depth == 3 and g.bands == (capi.BANDS_INTERLEAVED if bands == "interleaved" else capi.BANDS_CONTIGUOUS)
            with pytest.raises(capi.TrgError):
                g.set_option(capi.OPT_TIMING, 1)          # a pipelined group only enqueues
            for mode, root in ((capi.GATHER_ALL, 0), (capi.GATHER_ROOT, n - 1)):
                for L in (1, 2, 3, 4, 5, 8):
                    g.time_launches(True)
                    for i in range(L):
                        g.render(0, spps[i], bnc, gather=mode, root=root)
                    ref = refs[spps[L - 1]]
                    for r in (range(n) if mode == capi.GATHER_ALL else [root]):
                        assert np.array_equal(_bits(g.read_accum(r)), _bits(ref)), (n, mode, L, r)
                    for r in range(n):
                        ms = g.launch_ms(r)
                        assert len(ms) == L and all(t > 0 for t in ms), (n, r, ms)
                        assert g.launch_ms(r) == []
                    g.time_launches(False)
            # the run-ahead bound: at most two frames enqueued beyond what has finished
            for i in range(7):
                g.fence_wait(i % 2)
                g.render(0, spps[i], bnc, gather=capi.GATHER_ALL)
                g.fence_record(i % 2)
            g.fence_wait(0); g.fence_wait(1)
            for r in range(n):
                assert np.array_equal(_bits(g.read_accum(r)), _bits(refs[spps[6]])), (n, "fenced", r)
            g.reset_stats()
            g.render(0, 2, bnc, gather=capi.GATHER_NONE)
            g.render(0, 3, bnc, gather=capi.GATHER_NONE)
            one = g.stats().rays                           # (waits for every render stream of every rank)
            assert one > 0
            # back to one buffer: the accumulation is continued in place across launches
            g.set_pipeline(1)
            g.render(0, 2, bnc, gather=capi.GATHER_ALL)
            g.render(2, 3, bnc, gather=capi.GATHER_ALL)
            g.sync()
            for r in range(n):
                assert np.array_equal(_bits(g.read_accum(r)), _bits(refs[5])), (n, "progressive", r)
            pp = g.postprocess(0)                          # N1 on the group's frame (the unpacked image of an interleaved group)
            assert pp.shape == (h, w, 4) and pp[..., 3].min() == 255 and pp[..., :3].max() > 0
        finally:
            g.close()


def test_pipelined_group_exchange_through_rccl_on_one_device(capi, O, cornell, monkeypatch):
    """The RCCL side of the pipeline as far as ONE GPU can run it (TRG_GROUP_FORCE_RCCL: a communicator of one rank): the in-place
    ncclAllGather / the grouped send-recv go to the device's exchange stream behind the `rendered` event, the slot's next render waits
    for the `taken` event -- five frames over three slots without a host wait, the last one bit for bit."""
    monkeypatch.setenv("TRG_GROUP_FORCE_RCCL", "1")
    w, h, bnc = 120, 45, 3
    ref_ctx = make_ctx(O, cornell, w, h)
    try:
        ref_ctx.set_option(capi.OPT_STRICT, 1)
        ref_ctx.render(0, 2, bnc)
        ref = ref_ctx.read_accum()
    finally:
        ref_ctx.close()
    g = _group_ready(capi, O, cornell, [0], w, h)
    try:
        assert g.exchange == capi.EXCHANGE_RCCL and g.exchange_note == ""
        g.set_pipeline(3)
        for mode in (capi.GATHER_ALL, capi.GATHER_ROOT):
            for spp in (1, 3, 1, 3, 2):
                g.render(0, spp, bnc, gather=mode, root=0)
            g.fence_record(0); g.fence_wait(0)
            assert np.array_equal(_bits(g.read_accum(0)), _bits(ref)), mode
    finally:
        g.close()


def test_device_group_on_distinct_devices(capi, O, cornell, monkeypatch):
    """the cross-device calls -- hipMemcpyPeerAsync between two GPUs, stream waits on another device's events, and an RCCL
    communicator of more than one rank -- run only where there are two devices.  On such a machine: distinct devices, both exchanges,
    one buffer and pipelined, every gather mode, bit for bit against the plain context.  (Skipped on the single-GPU development boxes;
    the driver's multi-GPU node runs it.)"""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs two GPUs (this box has %d)" % ndev)
    w, h, bnc = 200, 93, 3
    ref_ctx = make_ctx(O, cornell, w, h)
    try:
        ref_ctx.set_option(capi.OPT_STRICT, 1)
        ref_ctx.render(0, 2, bnc)
        ref = ref_ctx.read_accum()
    finally:
        ref_ctx.close()
    for exchange in ("copy", "rccl"):
        monkeypatch.setenv("TRG_GROUP_EXCHANGE", exchange)
        for n in sorted({2, min(ndev, 8)}):
            g = _group_ready(capi, O, cornell, list(range(n)), w, h)
            try:
                assert g.exchange == (capi.EXCHANGE_COPY if exchange == "copy" else capi.EXCHANGE_RCCL)
                for depth in (1, 3):
                    g.set_pipeline(depth)
                    for mode, root in ((capi.GATHER_ALL, 0), (capi.GATHER_ROOT, n - 1)):
                        for spp in (1, 3, 1, 2):
                            g.render(0, spp, bnc, gather=mode, root=root)
                        for r in (range(n) if mode == capi.GATHER_ALL else [root]):
                            assert np.array_equal(_bits(g.read_accum(r)), _bits(ref)), (exchange, n, depth, mode, r)
                assert torch.cuda.current_device() == 0      # the group leaves the caller's device alone
            finally:
                g.close()


@pytest.mark.parametrize("schedule", ["lds", "lds_fp2", "lds_fp4", "lds_tail", "hbm", "hbm_regen", "hbm_fp2"])
def test_interleaved_bands_render_the_same_frame(capi, O, cornell, schedule):
    """trg_render_bands / trg_unpack_bands (round 4, SURVEY 8e's micro-bands): rank r of n renders the 8-row micro-bands r, r + n, ... of the
    image compactly into its slice of a bound buffer; all ranks' slices + the unpack give the plain context's frame BIT FOR BIT and the
    ray counts add up -- for every kernel a band can resolve to (frame-serial, frame lanes, tail compaction, HBM lock step, path
    regeneration), heights that are no multiple of 8, more ranks than micro-bands, and with the accumulation continued across launches."""
    import torch
    from toyraygun_amd.dist import unpack_bands_reference
    w, bnc = 150, (6 if schedule == "lds_tail" else 3)
    for h, n in ((93, 3), (93, 8), (40, 2), (21, 5)):
        c = make_ctx(O, cornell, w, h)
        try:
            c.set_option(capi.OPT_STRICT, 1)
            c.set_option(capi.OPT_FORCE_GLOBAL, 1 if schedule.startswith("hbm") else 0)
            c.set_option(capi.OPT_REGEN, 1 if schedule == "hbm_regen" else 0)
            c.set_option(capi.OPT_FRAME_SPLIT, {"lds_fp2": 2, "lds_fp4": 4, "hbm_fp2": 2}.get(schedule, 1))
            c.set_option(capi.OPT_TAIL_BOUNCE, 2 if schedule == "lds_tail" else 0)
            c.reset_stats()
            c.render(0, 2, bnc)
            c.render(2, 3, bnc)
            ref, rays = c.read_accum(), c.stats().rays
            with pytest.raises(capi.TrgError):
                c.render_bands(0, 1, bnc, n, n, 0)                     # rank out of range
            rows0, stride = capi.microband_rows(h, n, 0)
            if n * stride > h:
                with pytest.raises(capi.TrgError):
                    c.render_bands(0, 1, bnc, n, n - 1, (n - 1) * stride + 8)   # does not fit the context's own buffer
            compact = torch.zeros((n * stride, w, 4), dtype=torch.float32, device="cuda")
            image = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
            c.bind_accum(compact.data_ptr())
            c.reset_stats()
            for r in range(n):
                c.render_bands(0, 2, bnc, n, r, r * stride)
            for r in reversed(range(n)):                                # the continuation reads the compact band back
                c.render_bands(2, 3, bnc, n, r, r * stride)
            assert c.stats().rays == rays, (schedule, h, n)
            c.unpack_bands(compact.data_ptr(), image.data_ptr(), n)
            c.sync()
            got = image.cpu().numpy()
            assert np.array_equal(_bits(got), _bits(ref)), (schedule, h, n)
            assert np.array_equal(_bits(unpack_bands_reference(compact.cpu().numpy(), h, n)), _bits(ref))
            c.bind_accum(None)
            st = c.stats()
            assert st.last_regen == (1 if schedule == "hbm_regen" else 0) and st.scene_in_lds == (0 if schedule.startswith("hbm") else 1)
        finally:
            c.close()


def _checker_texture(n, cells, a, b, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n]
    chk = ((xx * cells // n) + (yy * cells // n)) % 2
    img = np.where(chk[..., None] == 0, np.array(a, np.uint8), np.array(b, np.uint8)).astype(np.uint8)
    img = np.clip(img.astype(int) + rng.integers(-20, 21, img.shape), 0, 255).astype(np.uint8)   # every texel distinct-ish
    return np.concatenate([img, np.full((n, n, 1), 255, np.uint8)], -1)


@pytest.mark.parametrize("nu,nv,in_lds", [(8, 5, 1), (40, 24, 0)])
def test_textured_scene_parity(capi, O, nu, nv, in_lds):
    """N4 (upstream to-do "OBJ and Texture support", README.md:18-22; Texture::loadFile, Texture.cpp:39-48): albedo textures.
    A textured sphere (texture coordinates from the product's Scene::addMesh, wrapped 3x around: repeat addressing) and a
    second texture on a quad, in the Cornell box.  Strict build: bit-exact against the oracle's restatement for all three
    schedules, LDS- and HBM-resident; shipped build within tolerance; the texture really is in the picture."""
    from toyraygun_amd import host
    v, n, col, tris = _uv_sphere(nu, nv, 0.33, (0.25, 1.1, 0.15))
    th = np.linspace(0.0, 1.0, nv + 1)
    ph = np.linspace(0.0, 3.0, nu, endpoint=False)           # u runs 0..3: wraps
    uv = np.array([[p, 1.0 - t * 2.0] for t in th for p in ph], np.float32)   # v runs 1..-1: negative coordinates too
    t1 = host.Texture(rgba=_checker_texture(64, 8, (230, 60, 40), (40, 90, 220), 1))
    t2 = host.Texture(rgba=_checker_texture(37, 5, (250, 250, 250), (30, 30, 30), 2))   # non-power-of-two
    hs = host.Scene.cornell_box()
    hs.add_textured_mesh(v, n, uv, tris, np.eye(4, dtype=np.float32), (0.9, 0.9, 0.9), 1, t1)
    qv = np.array([[-0.9, 0.02, -0.2], [-0.2, 0.02, -0.2], [-0.2, 0.02, 0.9], [-0.9, 0.02, 0.9]], np.float32)
    qn = np.tile(np.array([[0, 1, 0]], np.float32), (4, 1))
    quv = np.array([[0, 0], [2, 0], [2, 2], [0, 2]], np.float32)
    hs.add_textured_mesh(qv, qn, quv, [0, 2, 1, 0, 3, 2], np.eye(4, dtype=np.float32), (0.8, 0.8, 0.8), 1, t2)
    b = hs.buffers()
    uvs, ids, imgs = hs.texture_buffers()
    assert len(imgs) == 2 and (ids[:36] == 0).all() and (ids[36:36 + tris.shape[0]] == 1).all() and (ids[-2:] == 2).all()
    scene = O.OracleScene()
    scene.add_raw(b["positions"], b["normals"], b["colors"], b["material_ids"])
    w, h, spp, bnc = 160, 120, 3, 3
    off = O.pixel_offsets(w, h)
    c = capi.Context(w, h)
    try:
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        assert c.stats().scene_in_lds == in_lds
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets(off)
        c.set_option(capi.OPT_STRICT, 1)
        O.set_trig_mode(O.TRIG_PORTABLE)
        plain, _ = O.render(scene, w, h, spp, bnc, offsets=off)
        c.render(0, spp, bnc)
        assert np.array_equal(_bits(c.read_accum()), _bits(plain))       # no textures loaded yet
        scene.set_textures(uvs, ids, imgs)
        ref, rst = O.render(scene, w, h, spp, bnc, offsets=off)
        O.set_trig_mode(O.TRIG_LIBM)
        assert (np.abs(ref[..., :3] - plain[..., :3]).max(-1) > 1e-2).mean() > 0.03   # the textures are visible
        c.load_textures(uvs, ids, imgs)
        for k in KERNELS:
            c.set_option(capi.OPT_KERNEL, k)
            c.reset_stats()
            c.render(0, spp, bnc)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref)), "kernel %d" % k
            assert c.stats().rays == rst.rays
        ref_lib, _ = O.render(scene, w, h, spp, bnc, offsets=off)
        c.set_option(capi.OPT_STRICT, 0)
        c.set_option(capi.OPT_KERNEL, capi.KERNEL_AUTO)
        c.render(0, spp, bnc)
        rmse, frac_ok, worst = image_metrics(c.read_accum(), ref_lib)
        assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
        # the same scene and textures through a device group (trg_group_load_scene builds the host scene once, trg_group_load_textures)
        g = capi.Group([0], w, h)
        try:
            g.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
            g.load_textures(uvs, ids, imgs)
            g.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
            g.set_pixel_offsets_seed()
            g.set_option(capi.OPT_STRICT, 1)
            g.render(0, spp, bnc, gather=capi.GATHER_ALL)
            assert np.array_equal(_bits(g.read_accum(0)), _bits(ref)) and g.stats().rays == rst.rays
        finally:
            g.close()
        # removing the textures restores the untextured picture; a scene reload drops them too
        c.load_textures(uvs, ids, [])
        c.set_option(capi.OPT_STRICT, 1)
        c.render(0, spp, bnc)
        assert np.array_equal(_bits(c.read_accum()), _bits(plain))
        with pytest.raises(capi.TrgError):
            c.load_textures(uvs[:-3], ids[:-1], imgs)          # triangle count does not match the scene
        bad = ids.copy(); bad[0] = 3
        with pytest.raises(capi.TrgError):
            c.load_textures(uvs, bad, imgs)                    # names a texture that is not there
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


@pytest.mark.parametrize("k,bounces,spp", [(1, 4, 5), (3, 8, 37), (2, 15, 3), (7, 8, 18)])
def test_tail_compaction_is_bit_exact(capi, O, cornell, k, bounces, spp):
    """TRG_OPT_TAIL_BOUNCE (trg_tail.inc.h): the megakernel stops after bounce K-1, live paths are compacted into per-wavefront
    queues (ballot / prefix) and a second launch runs the deeper bounces with every lane live.  Same arithmetic per path in the same
    order: the strict build equals the oracle and the plain direct kernel bit for bit -- for frame counts that do not fill the
    16-frame chunks, for a launch continued from an earlier average, for row bands, with textures absent -- and counts the same rays."""
    w, h = 200, 150
    off = O.pixel_offsets(w, h)
    O.set_trig_mode(O.TRIG_PORTABLE)
    ref, rst = O.render(cornell, w, h, spp, bounces, offsets=off)
    O.set_trig_mode(O.TRIG_LIBM)
    c = make_ctx(O, cornell, w, h, offsets=off)
    try:
        c.set_option(capi.OPT_STRICT, 1)
        c.set_option(capi.OPT_TAIL_BOUNCE, k)
        c.reset_stats()
        c.render(0, spp, bounces)
        st = c.stats()
        assert np.array_equal(_bits(c.read_accum()), _bits(ref)) and st.rays == rst.rays and st.shaded_hits == rst.shaded_hits
        # continuation + bands: frames [0, a) on the whole image, then [a, spp) in three bands
        a = max(1, spp // 3)
        c.render(0, a, bounces)
        for r0, n in ((0, 50), (50, 63), (113, 37)):
            c.render(a, spp - a, bounces, r0, n)
        assert np.array_equal(_bits(c.read_accum()), _bits(ref))
        # TRG_OPT_TAIL_SORT (round 4): a tile's queued paths sorted by direction octant (+ origin cell) before every tail launch -- the
        # order of the paths changes, the arithmetic of a path does not
        for mode in (1, 2, 3):
            c.set_option(capi.OPT_TAIL_SORT, mode)
            c.reset_stats()
            c.render(0, spp, bounces)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref)) and c.stats().rays == rst.rays, ("sorted", mode)
        with pytest.raises(capi.TrgError):
            c.set_option(capi.OPT_TAIL_SORT, 4)
        c.set_option(capi.OPT_TAIL_SORT, 0)
        # TRG_OPT_TAIL_REFILL (round 5): one bounce per tail launch, the lanes of a wavefront take queue entries at their own pace (render_rtail_kernel):
        # the same per-path arithmetic -- whole frame, continued average, row bands, ray counts
        if not capi.has_experiments():    # (the refilling tail lost to the lock-step one and lives in the experiments library: the product refuses it)
            with pytest.raises(capi.TrgError, match="experiment"):
                c.set_option(capi.OPT_TAIL_REFILL, 1)
        for refill in ((1, 0) if capi.has_experiments() else (0,)):
            c.set_option(capi.OPT_TAIL_REFILL, refill)
            c.reset_stats()
            c.render(0, spp, bounces)
            st = c.stats()
            assert np.array_equal(_bits(c.read_accum()), _bits(ref)) and st.rays == rst.rays and st.shaded_hits == rst.shaded_hits, ("refill", refill)
            c.render(0, a, bounces)
            for r0, n in ((0, 50), (50, 63), (113, 37)):
                c.render(a, spp - a, bounces, r0, n)
            assert np.array_equal(_bits(c.read_accum()), _bits(ref)), ("refill bands", refill)
        with pytest.raises(capi.TrgError):
            c.set_option(capi.OPT_TAIL_REFILL, 2)
        c.set_option(capi.OPT_TAIL_REFILL, -1)
        # off = the plain kernel; auto picks it for deep paths only
        c.set_option(capi.OPT_TAIL_BOUNCE, 0)
        c.render(0, spp, bounces)
        assert np.array_equal(_bits(c.read_accum()), _bits(ref))
        c.set_option(capi.OPT_TAIL_BOUNCE, -1)
        c.render(0, spp, bounces)
        assert np.array_equal(_bits(c.read_accum()), _bits(ref))
        # shipped build: within tolerance of the libm oracle
        ref_lib, _ = O.render(cornell, w, h, spp, bounces, offsets=off)
        c.set_option(capi.OPT_STRICT, 0)
        c.set_option(capi.OPT_TAIL_BOUNCE, k)
        c.render(0, spp, bounces)
        rmse, frac_ok, worst = image_metrics(c.read_accum(), ref_lib)
        assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok, worst)
        with pytest.raises(capi.TrgError):
            c.set_option(capi.OPT_TAIL_BOUNCE, 99)
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
        c.close()


def test_async_frames_keep_their_uniforms(capi, O, cornell):
    """The asynchronous renderFrame() lets frames wait for a later launch while the device is busy.  Frames accepted before a
    camera move must still be rendered with the camera they were requested with (setCameraPosition is a non-virtual base-class
    setter, so the backend compares uniform snapshots): the result equals two explicit launches through the C ABI, bit for bit."""
    from toyraygun_amd import host
    w, h, fa, fb, bnc = 640, 480, 9, 7, 3
    eye_b = (0.3, 1.1, 3.0)
    got, launches = host.async_camera_move(w, h, fa, fb, eye_b, bnc)
    assert 2 <= launches <= fa + fb
    c = capi.Context(w, h)
    try:
        b = host.Scene.cornell_box().buffers()
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_pixel_offsets_seed()
        c.set_uniforms(host.uniforms(w, h)[0])
        c.render(0, fa, bnc)
        c.set_uniforms(host.uniforms(w, h, eye=eye_b)[0])
        c.render(fa, fb, bnc)
        assert np.array_equal(_bits(got), _bits(c.read_accum()))
        # and it differs from rendering all frames with either camera
        c.render(0, fa + fb, bnc)
        assert not np.array_equal(_bits(got), _bits(c.read_accum()))
    finally:
        c.close()


def test_plugin_on_a_device_group(capi, O, monkeypatch):
    """HipRenderer::setDevices: the plugin on a device group (trg_group_*), also a group of one -- the frame loop then goes through
    trg_group_fence_wait / trg_group_render / trg_group_fence_record, the scene through trg_group_load_scene and the textures
    through trg_group_load_textures.  Same frames as the plain renderer bit for bit: the asynchronous loop with a camera move in
    the middle (coalesced launches, three in flight), an HBM scene built on the device, and a textured scene."""
    from toyraygun_amd import host
    w, h, fa, fb, bnc = 320, 240, 9, 7, 3
    eye_b = (0.3, 1.1, 3.0)
    plain, launches_plain = host.async_camera_move(w, h, fa, fb, eye_b, bnc)
    grouped, launches = host.async_camera_move(w, h, fa, fb, eye_b, bnc, devices=[0])
    assert 2 <= launches <= fa + fb
    assert np.array_equal(_bits(grouped), _bits(plain))
    lattice = host.Scene.cornell_lattice(12)     # 20,772 triangles: lives in HBM
    for builder in (0, 1):      # (each against the plain renderer with the SAME builder: only the host builder makes box leaves of the cubes)
        ref, _ = host.render_scene(lattice, 160, 120, 4, 3, device_build=builder)
        got, _ = host.render_scene(lattice, 160, 120, 4, 3, device_build=builder, devices=[0])
        assert np.array_equal(_bits(got), _bits(ref)), builder
    v, n, col, tris = _uv_sphere(12, 8, 0.33, (0.25, 1.1, 0.15))
    uv = np.array([[p, t] for t in np.linspace(0.0, 1.0, 9) for p in np.linspace(0.0, 3.0, 12, endpoint=False)], np.float32)
    tex = host.Texture(rgba=_checker_texture(64, 8, (230, 60, 40), (40, 90, 220), 1))
    hs = host.Scene.cornell_box()
    hs.add_textured_mesh(v, n, uv, tris, np.eye(4, dtype=np.float32), (0.9, 0.9, 0.9), 1, tex)
    a, _ = host.render_scene(hs, 160, 120, 3, 3)
    b, _ = host.render_scene(hs, 160, 120, 3, 3, devices=[0])
    assert np.array_equal(_bits(a), _bits(b))
    untextured, _ = host.render_scene(host.Scene.cornell_box(), 160, 120, 3, 3, devices=[0])
    assert not np.array_equal(_bits(b), _bits(untextured))
    with pytest.raises(RuntimeError):
        host.render_scene(hs, 160, 120, 1, 3, devices=[0, 0])     # a device listed twice: init() fails, nothing is rendered
    # ... unless the bands travel by copies (TRG_GROUP_EXCHANGE=copy): then three contexts on this one device run the plugin's n > 1 path
    # -- bands gathered on the first context, which reads back -- for the asynchronous loop, the device-built HBM scene and the textures
    monkeypatch.setenv("TRG_GROUP_EXCHANGE", "copy")
    grouped3, _ = host.async_camera_move(w, h, fa, fb, eye_b, bnc, devices=[0, 0, 0])
    assert np.array_equal(_bits(grouped3), _bits(plain))
    got, _ = host.render_scene(lattice, 160, 120, 4, 3, device_build=1, devices=[0, 0])
    assert np.array_equal(_bits(got), _bits(ref))
    b3, _ = host.render_scene(hs, 160, 120, 3, 3, devices=[0, 0, 0])
    assert np.array_equal(_bits(a), _bits(b3))


# ------------------------------------------------------------------ round 3: tile order, leaf records
@pytest.mark.parametrize("force_global", [0, 1])
def test_tile_order_never_changes_the_image(capi, O, cornell, force_global):
    """TRG_OPT_TILE_ORDER: which tile a workgroup renders is a question of time only.  The column order and every XCD-aware layout
    (1, 2, 4, 8 column strips x 8, 4, 2, 1 row bands; the launch is padded to eight equal regions) give the strict image and the ray
    counts of the oracle bit for bit -- partial tiles, fewer tile rows than bands, row bands of a frame, the frame-split, tail-compaction
    and regeneration schedules included."""
    for (w, h, spp, bounces) in ((97, 45, 3, 3), (200, 24, 4, 2), (16, 130, 2, 5), (64, 64, 5, 8)):
        off = O.pixel_offsets(w, h)
        O.set_trig_mode(O.TRIG_PORTABLE)
        try:
            ref, rst = O.render(cornell, w, h, spp, bounces, offsets=off)
        finally:
            O.set_trig_mode(O.TRIG_LIBM)
        c = make_ctx(O, cornell, w, h, offsets=off)
        try:
            c.set_option(capi.OPT_FORCE_GLOBAL, force_global)
            c.set_option(capi.OPT_STRICT, 1)
            for order in (0, 1, 2, 4, 8, 17, 18, 20, 24, 32, 66, 72, -1):
                c.set_option(capi.OPT_TILE_ORDER, order)
                for fsplit, regen in ((1, 0), (2, 0), (4, 1), (1, 1)):
                    c.set_option(capi.OPT_FRAME_SPLIT, fsplit)
                    c.set_option(capi.OPT_REGEN, regen)
                    c.reset_stats()
                    c.render(0, spp, bounces)
                    st = c.stats()
                    assert st.last_tile_order == (max(order, 0) if order < 64 or st.last_regen else order - 64)   # automatic = the column order; 64 + n: the queues exist in the regeneration kernel only
                    assert np.array_equal(_bits(c.read_accum()), _bits(ref)) and st.rays == rst.rays, (w, h, order, fsplit, regen)
                # two row bands of the frame
                c.set_option(capi.OPT_FRAME_SPLIT, 0); c.set_option(capi.OPT_REGEN, -1)
                c.render(0, spp, bounces, 0, h // 3); c.render(0, spp, bounces, h // 3, h - h // 3)
                assert np.array_equal(_bits(c.read_accum()), _bits(ref)), (w, h, order, "bands")
        finally:
            c.close()


@pytest.mark.parametrize("builder", [0, 2])
def test_stack_levels_in_lds_never_change_the_image(capi, O, builder):
    """TRG_OPT_STACK_LDS_LEVELS: an HBM-resident scene keeps the first levels of a thread's traversal stack in LDS and the deeper ones in a
    per-launch scratch in memory; where the line is drawn is a question of time only.  With 2 (everything but the sentinel in memory),
    3, 5 and the default 12 levels in LDS a 3,000-triangle soup (host SAH and the deeper LBVH tree) gives the oracle's strict image and
    ray counts bit for bit in the lock-step, frame-lane, regeneration and wavefront schedules -- the scratch path is otherwise only
    reached by the few rays that hold more than eleven pending nodes."""
    soup = _random_soup(O, 3000, 5)
    w, h, spp, bnc = 96, 64, 3, 4
    off = O.pixel_offsets(w, h)
    O.set_trig_mode(O.TRIG_PORTABLE)
    try:
        ref, rst = O.render(soup, w, h, spp, bnc, offsets=off)
    finally:
        O.set_trig_mode(O.TRIG_LIBM)
    b = soup.buffers()
    c = capi.Context(w, h)
    try:
        c.set_option(capi.OPT_GPU_BUILD, builder)
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        assert c.stats().scene_in_lds == 0
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets(off)
        c.set_option(capi.OPT_STRICT, 1)
        for levels in (2, 3, 5, 12):
            c.set_option(capi.OPT_STACK_LDS_LEVELS, levels)
            for kernel, fsplit, regen in ((capi.KERNEL_DIRECT, 1, 0), (capi.KERNEL_DIRECT, 2, 0), (capi.KERNEL_DIRECT, 1, 1), (capi.KERNEL_DIRECT, 2, 1)) + (((capi.KERNEL_WAVEFRONT, 1, 0),) if capi.has_experiments() else ()):
                c.set_option(capi.OPT_KERNEL, kernel); c.set_option(capi.OPT_FRAME_SPLIT, fsplit); c.set_option(capi.OPT_REGEN, regen)
                c.reset_stats()
                c.render(0, spp, bnc)
                assert np.array_equal(_bits(c.read_accum()), _bits(ref)) and c.stats().rays == rst.rays, (levels, kernel, fsplit, regen)
        for bad in (1, 13):
            with pytest.raises(capi.TrgError):
                c.set_option(capi.OPT_STACK_LDS_LEVELS, bad)
    finally:
        c.close()


def test_workgroups_are_dealt_round_robin_over_the_xcds(capi, O, cornell):
    """What the XCD-aware tile orders rely on for SPEED (never for results, and the default order does not use it): workgroup b of a launch
    runs on XCD (b + k) % 8 for one k per launch (MI355X_MICROARCH.md, workgroup dispatch).  Read back from HW_REG_XCC_ID.  The ids must
    be valid; a machine that places workgroups differently (another partition mode) is reported as a skip with the share found, not as a
    failure -- there the XCD-aware orders merely lose their point."""
    c = make_ctx(O, cornell, 64, 64)
    try:
        ids = c.xcc_ids(4096)
        assert ids.max() <= 7
        k = int(ids[0]) % 8
        share = float((ids == (np.arange(4096) + k) % 8).mean())
        if len(set(ids.tolist())) != 8 or share < 0.98:
            pytest.skip("workgroups are not dealt round-robin over 8 XCDs here (%d XCDs seen, %.1f %% on XCD (b + k) %% 8)" % (len(set(ids.tolist())), 100.0 * share))
    finally:
        c.close()


def test_distance_ties_in_leaf_records_go_to_the_lower_original_index(capi, O):
    """An HBM-resident scene names a hit by its 128-byte leaf RECORD; the contract's tie rule (equal distance -> lower original
    primitive index) is kept by reading the held record's original index back on a tie.  Every triangle of the soup twice, the copies
    in shuffled order: nearest hits report the lower of the two indices, exactly as the oracle's brute force does, for every builder."""
    rng = np.random.default_rng(77)
    bb = O.OracleScene.cornell_box().buffers()
    nt = bb["material_ids"].shape[0]
    perm = rng.permutation(2 * nt)
    def dup(a):   # [3 * nt, 3] per-corner attribute -> both copies of every triangle, shuffled
        return np.concatenate([a, a]).reshape(2 * nt, 3, 3)[perm].reshape(-1, 3)
    pos, nrm, col = dup(bb["positions"]), dup(bb["normals"]), dup(bb["colors"])
    mat = np.concatenate([bb["material_ids"], bb["material_ids"]])[perm]
    idx = np.arange(pos.shape[0], dtype=np.uint32)
    scene = O.OracleScene()
    scene.add_raw(pos, nrm, col, mat)
    rays = _rays(O, 20000, 5)
    ref = O.intersect_nearest(scene, rays, brute=True)
    hit = ref["primitiveIndex"] >= 0
    assert hit.sum() > 5000
    twin = np.empty(2 * nt, np.int64); inv = np.argsort(perm)
    twin[:] = inv[(perm + nt) % (2 * nt)]   # the other copy of triangle k
    assert (ref["primitiveIndex"][hit] < twin[ref["primitiveIndex"][hit]]).all()   # the oracle itself reports the lower copy
    for builder in (0, 1, 2, 3):
        c = capi.Context(32, 32)
        try:
            c.set_option(capi.OPT_FORCE_GLOBAL, 1)
            c.set_option(capi.OPT_GPU_BUILD, builder)
            c.load_scene(pos, nrm, col, idx, mat)
            for strict in (1, 0):
                c.set_option(capi.OPT_STRICT, strict)
                got = c.trace(rays)
                if strict:
                    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), builder
                else:   # the shipped build may round a distance differently, never the tie rule: a reported triangle is the lower copy
                    g = got["primitiveIndex"]
                    assert (g[g >= 0] < twin[g[g >= 0]]).all() and (g == ref["primitiveIndex"]).mean() > 0.999, builder
        finally:
            c.close()


def test_trace_ray_masks_beyond_the_two_low_bits(capi, O):
    """the shipped build's plane test on an LDS-resident scene reads its mask off a u16 per record, (original index << 2) |
    (material id & 3).  A ray mask with bits >= 2 set (0xFFFFFFFF, 4, 0x14 ...) must not match primitive-INDEX bits, and a material id
    of 4 or more must still be honoured by such a ray: trg_trace routes those rays through the HBM records, which keep the whole id
    (MetalRenderer.mm:269,276: ray.mask & triangle mask).  Material id 0 is hit by no ray at all.  Strict build: the oracle's records
    bit for bit; shipped build: the same primitive except on grazing rays."""
    rng = np.random.default_rng(505)
    n = 60
    ctr = rng.uniform(-0.8, 0.8, (n, 3))
    tri = (ctr[:, None, :] + rng.normal(0, 0.25, (n, 3, 3))).astype(np.float32)
    mats = rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 0x10, 0x80000000], n)
    mats[:10] = [0, 0, 4, 4, 8, 1, 2, 3, 0x10, 5]
    s = O.OracleScene()
    eye = np.eye(4, dtype=np.float32)
    for k in range(n):
        s.add_geometry(tri[k], [0, 1, 2], eye, (0.5, 0.5, 0.5), int(mats[k]))
    b = s.buffers()
    m = 20000
    rays = np.zeros(m, O.RAY_DTYPE)
    rays["origin"] = rng.uniform(-1.5, 1.5, (m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3))
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["maxDistance"] = np.inf
    c = capi.Context(16, 16)
    try:
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        assert c.stats().scene_in_lds == 1
        for masks in ([0xFFFFFFFF], [4], [0x14, 8, 0x80000000], [1, 3], [0], [1, 2, 3, 4, 0xFFFFFFFF, 0xC]):
            rays["mask"] = rng.choice(np.array(masks, np.uint64), m).astype(np.uint32)
            ref = O.intersect_nearest(s, rays)
            ref_any = O.intersect_any(s, rays) >= 0
            hitm = np.asarray(b["material_ids"], np.uint32)[np.maximum(ref["primitiveIndex"], 0)]
            assert ((hitm & rays["mask"]) != 0)[ref["primitiveIndex"] >= 0].all()          # the oracle itself honours the 32-bit mask
            if masks == [0]:
                assert (ref["primitiveIndex"] == -1).all()
            for strict in (1, 0):
                c.set_option(capi.OPT_STRICT, strict)
                got = c.trace(rays)
                got_any = c.trace(rays, any_hit=True) >= 0
                if strict:
                    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), masks
                    assert np.array_equal(got_any, ref_any), masks
                else:
                    diff = got["primitiveIndex"] != ref["primitiveIndex"]
                    if diff.any():
                        _, _, margin = O.nearest_f64(s, rays[diff])
                        assert (margin < 1e-5).all(), (masks, int(diff.sum()))
                    assert diff.mean() < 3e-3, (masks, diff.mean())
                    assert (got_any != ref_any).mean() < 2e-3, masks
                    hit = got["primitiveIndex"] >= 0
                    gm = np.asarray(b["material_ids"], np.uint32)[np.maximum(got["primitiveIndex"], 0)]
                    assert ((gm & rays["mask"]) != 0)[hit].all(), masks                     # never a masked-out triangle, never material id 0
    finally:
        c.close()


def test_pipelined_interleaved_copy_exchange_every_step(capi, O, cornell, monkeypatch):
    """ADVICE r04 (write-after-read on slot re-use): with interleaved bands a destination's exchange stream UNPACKS its own band of
    frame[r][slot] too, so the render that re-uses the slot must also wait for taken[r][slot] -- and a gather to a root must make the root wait
    for its own unpack.  Big frames and next-to-free renders (1 spp, 1 bounce at 1280 x 720: the unpack of step i is still running when
    step i + depth wants the slot), depth 2, three contexts on one device; EVERY step's image is checked -- sequence length L ends on step
    L -- against the plain context, on every rank that holds it.  After a gather to a root the other ranks still show their OWN micro-bands
    (trg.h: read_accum returns the image either way)."""
    monkeypatch.setenv("TRG_GROUP_EXCHANGE", "copy")
    w, h, n = 1280, 720, 3
    ref_ctx = make_ctx(O, cornell, w, h)
    refs = {}
    try:
        ref_ctx.set_option(capi.OPT_STRICT, 1)
        for bnc in (1, 2):
            ref_ctx.render(0, 1, bnc)
            refs[bnc] = ref_ctx.read_accum()
    finally:
        ref_ctx.close()
    assert not np.array_equal(refs[1], refs[2])
    own = [np.array([y for y in range(h) if (y // 8) % n == r]) for r in range(n)]    # image rows of rank r's micro-bands
    g = _group_ready(capi, O, cornell, [0] * n, w, h)
    try:
        g.set_bands(capi.BANDS_INTERLEAVED)
        g.set_pipeline(2)
        for mode, root in ((capi.GATHER_ALL, 0), (capi.GATHER_ROOT, 1)):
            for L in range(1, 8):
                for i in range(L):
                    g.render(0, 1, 1 + (i & 1), gather=mode, root=root)      # consecutive steps are DIFFERENT pictures
                ref = refs[1 + ((L - 1) & 1)]
                for r in range(n):
                    img = g.read_accum(r)
                    if mode == capi.GATHER_ALL or r == root:
                        assert np.array_equal(_bits(img), _bits(ref)), (mode, L, r)
                    else:
                        assert np.array_equal(_bits(img[own[r]]), _bits(ref[own[r]])), (mode, L, r, "own band")
        # reconfiguring a drained group must not use up the context's per-stream scratch slots (16 slots, `depth` new streams per call)
        for k in range(8):
            g.set_pipeline(8 if k & 1 else 7)
            g.render(0, 1, 1, gather=capi.GATHER_ALL)
        g.sync()
        assert np.array_equal(_bits(g.read_accum(0)), _bits(refs[1]))
    finally:
        g.close()
