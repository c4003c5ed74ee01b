"""CPU tests of the product's host side: the C ABI library loads and exports what include/trg.h declares,
the C++ plugin surface (Engine/Renderer/Scene/Shader mirror) reproduces the oracle's scene, camera,
uniforms and offsets, and the BVH builder is sound.  No compute calls: there is no GPU here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol(built):
    from toyraygun_amd import capi
    hdr = open(os.path.join(ROOT, "include", "trg.h")).read()
    declared = sorted(set(re.findall(r"\b(trg_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = capi.load()
    for name in declared:
        assert hasattr(L, name), "libtoyraygun_hip.so does not export %s" % name
    assert sorted(capi.SYMBOL_NAMES) == declared       # the Python binding covers the whole header
    out = subprocess.run(["nm", "-D", "--defined-only", capi.HIP_SO], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(r"\bT %s\b" % name, out), name


def test_scene_blob_layout_is_64_bit_and_refuses_what_cannot_fit(built):
    """the blob offsets used to be summed in uint32 and wrapped from ~30 M triangles on, so the 4 GiB check
    passed on a wrapped total.  The layout is now planned in 64 bits and refused before any allocation."""
    import ctypes as C
    from toyraygun_amd import capi
    L = capi.load()
    total = C.c_uint64()
    off = (C.c_uint32 * 7)()
    assert L.trg_debug_scene_layout(1_022_244, 246_000, C.byref(total), off) == capi.OK
    o = list(off)
    # a scene this size has no LDS part: the quantised wide nodes lead the blob, the 128-byte leaf records follow on their own lines
    assert o[:6] == [0] * 6 and o[6] % 128 == 0 and o[6] >= 246_000 * 64
    assert total.value >= o[6] + 1_022_244 * 256       # two sets of leaf records since round 4: Moeller-Trumbore rows (strict build) and planes (shipped build)
    # 14 M triangles fit (2 x 128 B per triangle + nodes = 3.8e9); 16 M (4.35e9) would not
    assert L.trg_debug_scene_layout(14_000_000, 3_500_000, C.byref(total), None) == capi.OK
    assert total.value > 14_000_000 * 256
    assert L.trg_debug_scene_layout(16_000_000, 4_000_000, C.byref(total), None) == capi.ERR_RANGE
    assert L.trg_debug_scene_layout(30_000_000, 7_500_000, C.byref(total), None) == capi.ERR_RANGE
    # 50 M triangles: the old uint32 sums wrapped to ~2.9e9 and passed; now refused with the true size
    assert L.trg_debug_scene_layout(50_000_000, 12_000_000, C.byref(total), None) == capi.ERR_RANGE
    assert total.value > 2**32
    assert L.trg_debug_scene_layout(40_000_000, 0, C.byref(total), None) == capi.ERR_RANGE  # even without a single node


def test_no_gpu_means_loud_failure(built):
    import torch
    from toyraygun_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.TrgError) as e:
        capi.Context(16, 16)
    assert e.value.code == capi.ERR_NODEV and "no CPU fallback" in str(e.value)
    with pytest.raises(capi.TrgError):
        capi.Context(0, 16)


def test_plain_c_consumer(built, tmp_path):
    """include/trg.h compiles as C (gcc -std=c99 -pedantic -Werror), the library links from a C program, and on a box
    without a GPU that program gets TRG_ERR_NODEV with the "no CPU fallback" message from trg_create."""
    from toyraygun_amd import capi
    exe = str(tmp_path / "c_consumer")
    lib_dir = os.path.dirname(capi.HIP_SO)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "helpers", "c_consumer.c"), "-o", exe,
                           "-L" + lib_dir, "-ltoyraygun_hip", "-Wl,-rpath," + lib_dir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "no CPU fallback" in r.stdout or "gpu path ok" in r.stdout


def test_struct_layouts(built):
    from toyraygun_amd import capi
    assert C.sizeof(capi.Uniforms) == 176                     # Uniforms.h:19-41 / MetalRenderer.mm:34-36
    assert capi.Uniforms.cam_pos.offset == 16 and capi.Uniforms.inv_view_proj.offset == 32
    assert capi.Uniforms.light_pos.offset == 96 and capi.Uniforms.light_color.offset == 160
    assert capi.RAY_DTYPE.itemsize == 48 and capi.RAY_DTYPE.fields["direction"][1] == 16 and capi.RAY_DTYPE.fields["color"][1] == 32
    assert capi.ISECT_DTYPE.itemsize == 16


def test_python_binding_matches_the_header(built, tmp_path):
    """The ctypes binding (toyraygun_amd/capi.py) restates enums and the trg_stats layout of include/trg.h by hand: a C probe compiled
    against the header prints the real values, and every one the binding names must agree."""
    import re
    import subprocess
    from toyraygun_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "trg.h")).read()
    names = sorted(set(re.findall(r"\b(TRG_(?:OPT|KERNEL|GATHER|ERR)_[A-Z_0-9]+)\b", hdr)))
    fields = re.search(r"typedef struct trg_stats \{(.*?)\} trg_stats;", hdr, re.S).group(1)
    fields = re.sub(r"/\*.*?\*/", "", fields, flags=re.S)
    members = [m for decl in re.findall(r"(?:uint64_t|uint32_t|double)\s+([^;]+);", fields) for m in re.split(r"\s*,\s*", decl.strip())]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "trg.h"', 'int main(void) {']
    src += ['  printf("%s %%d\\n", (int)%s);' % (n, n) for n in names]
    src += ['  printf("sizeof_stats %d\\n", (int)sizeof(trg_stats));']
    src += ['  printf("off_%s %%d\\n", (int)offsetof(trg_stats, %s));' % (m, m) for m in members]
    src += ['  return 0; }']
    c_file = tmp_path / "probe.c"
    c_file.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(c_file), "-o", str(exe)])
    vals = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:   # e.g. TRG_OPT_REGEN <-> capi.OPT_REGEN, TRG_ERR_RANGE <-> capi.ERR_RANGE
        py = n[4:]
        if hasattr(capi, py):
            assert getattr(capi, py) == int(vals[n]), n
    for need in ("OPT_REGEN", "OPT_GPU_BUILD", "OPT_TAIL_BOUNCE", "KERNEL_WAVEFRONT", "GATHER_ROOT"):
        assert hasattr(capi, need)
    assert C.sizeof(capi.Stats) == int(vals["sizeof_stats"])
    py_fields = dict((name, getattr(capi.Stats, name).offset) for name, _ in capi.Stats._fields_)
    assert set(py_fields) == set(members)
    for m in members:
        assert py_fields[m] == int(vals["off_" + m]), m


def test_host_scene_equals_oracle_scene(built, O):
    from toyraygun_amd import host
    hs, os_ = host.Scene.cornell_box().buffers(), O.OracleScene.cornell_box().buffers()
    for k in hs:
        assert hs[k].dtype == os_[k].dtype and np.array_equal(hs[k], os_[k]), k
    # the public builders one by one, with an arbitrary transform
    m = host.mtx_srt((0.5, 2.0, 1.5), (0.2, -1.3, 0.7), (0.1, 0.2, -0.3))
    a, b = host.Scene(), O.OracleScene()
    for kind, col in (("cube", (0.1, 0.2, 0.3)), ("plane", (0.9, 0.8, 0.7)), ("light", (1, 1, 1))):
        a.add(kind, col, m)
        b.add(kind, col, m)
    ha, hb = a.buffers(), b.buffers()
    for k in ha:
        assert np.array_equal(ha[k], hb[k]), k
    assert ha["material_ids"].tolist() == [1] * 14 + [2] * 2
    # the C4 lattice scene (product builder == oracle builder)
    la, lb = host.Scene.cornell_lattice(5).buffers(), O.OracleScene.cornell_lattice(5).buffers()
    assert la["material_ids"].shape == (36 + 12 * 125,)
    for k in la:
        assert np.array_equal(la[k], lb[k]), k


def test_host_matrices_uniforms_offsets_equal_oracle(built, O):
    from toyraygun_amd import host
    fp = C.POINTER(C.c_float)
    for args in (((0.6, 0.6, 0.6), (0.0, 0.3, 0.0), (0.3275, 0.3, 0.3725)), ((2, 2, 2), (0, 0, np.pi), (0, 1, 0)),
                 ((1, 2, 3), (-0.4, 2.2, 0.9), (4, 5, 6))):
        ref = np.zeros(16, np.float32)
        O.lib().orc_mtx_srt(ref.ctypes.data_as(fp), *[float(v) for t in args for v in t])
        assert np.array_equal(host.mtx_srt(*args), ref)
    for (w, h) in ((256, 256), (1024, 768), (1920, 1080), (3840, 2160)):
        for f in (0, 7):
            u, vp = host.uniforms(w, h, f)
            assert bytes(u) == O.uniforms_bytes(O.make_uniforms(w, h, f))
        assert np.array_equal(host.mtx_inverse(vp).reshape(4, 4).T.reshape(-1), np.array(u.inv_view_proj, np.float32))
    assert np.array_equal(host.random_texture(40, 24), O.pixel_offsets(40, 24))
    assert np.array_equal(host.random_texture(8, 8, 1234), O.pixel_offsets(8, 8, 1234))


def test_add_mesh_smooth_normals(built):
    from toyraygun_amd import host
    s = host.Scene()
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    n = np.array([[0, 0, 1], [0.6, 0, 0.8], [0, 0.6, 0.8]], np.float32)
    s.add_mesh(v, n, [0, 1, 2], np.eye(4, dtype=np.float32), (0.5, 0.5, 0.5), 1)
    b = s.buffers()
    assert b["positions"].shape == (3, 3) and np.allclose(b["normals"], n) and b["material_ids"].tolist() == [1]


def test_obj_loader(built, O, tmp_path):
    """Scene::addObj: a cube written as OBJ quads gives the same triangle soup as addCube up to
    the fan triangulation; smooth normals survive; negative indices and v/vt/vn corners parse."""
    from toyraygun_amd import host
    p = tmp_path / "cube.obj"
    v = [(-.5, -.5, -.5), (.5, -.5, -.5), (-.5, .5, -.5), (.5, .5, -.5), (-.5, -.5, .5), (.5, -.5, .5), (-.5, .5, .5), (.5, .5, .5)]
    quads = [(0, 4, 6, 2), (1, 3, 7, 5), (0, 1, 5, 4), (2, 6, 7, 3), (0, 2, 3, 1), (4, 5, 7, 6)]   # the faces of Scene::addCube
    with open(p, "w") as f:
        f.write("# cube\n")
        for x in v:
            f.write("v %g %g %g\n" % x)
        for q in quads:
            f.write("f " + " ".join(str(i + 1) for i in q) + "\n")
    m = host.mtx_srt((0.6, 1.2, 0.6), (0, -0.3, 0), (-0.335, 0.6, -0.29))
    a, b = host.Scene(), host.Scene()
    assert a.add_obj(p, m, (0.7, 0.7, 0.7)) == 12
    b.add("cube", (0.7, 0.7, 0.7), m)
    A, B = a.buffers(), b.buffers()
    assert np.array_equal(A["positions"], B["positions"]) and np.array_equal(A["normals"], B["normals"])
    assert np.array_equal(A["indices"], np.arange(36, dtype=np.uint32)) and A["material_ids"].tolist() == [1] * 12
    # normals, texture coordinates and negative (relative) indices
    p2 = tmp_path / "tri.obj"
    p2.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nvn 0.6 0 0.8\nf 1/1/1 2/1/2 3/1/1\nf -3//1 -2//1 -1//1\n")
    c = host.Scene()
    assert c.add_obj(p2, np.eye(4, dtype=np.float32), (1, 1, 1), 2) == 2
    Cb = c.buffers()
    assert np.allclose(Cb["normals"][:3], [[0, 0, 1], [0.6, 0, 0.8], [0, 0, 1]]) and Cb["material_ids"].tolist() == [2, 2]
    assert np.array_equal(Cb["positions"][3:], Cb["positions"][:3])
    assert host.Scene().add_obj(tmp_path / "missing.obj", np.eye(4, dtype=np.float32), (1, 1, 1)) == -1


def _walk(nodes, tris, ntris):
    """Check the flattened BVH: every triangle is in exactly one leaf and inside every box above it."""
    seen = np.zeros(ntris, np.int32)
    depth_max = 0
    stack = [(0, 0, np.full(3, -np.inf), np.full(3, np.inf))]
    child_bits = nodes[:, 12:14].copy().view(np.int32)
    while stack:
        ni, d, plo, phi = stack.pop()
        n = nodes[ni]
        boxes = [(np.array([n[0], n[2], n[8]]), np.array([n[1], n[3], n[9]])),
                 (np.array([n[4], n[6], n[10]]), np.array([n[5], n[7], n[11]]))]
        for c in range(2):
            lo, hi = boxes[c]
            ref = int(child_bits[ni, c])
            if ref >= 0:
                stack.append((ref, d + 1, lo, hi))
            else:
                code = ~ref
                first, count = code >> 3, (2 if (code & 7) == 7 else (code & 7) + 1)     # (7: a quad leaf -- two triangles of a parallelogram, bvh_build.h)
                depth_max = max(depth_max, d + 1)
                for k in range(first, first + count):
                    rec = tris[k]
                    prim = int(rec[3:4].view(np.int32)[0])
                    v0, e1, e2 = rec[0:3], rec[4:7], rec[8:11]
                    for p in (v0, v0 + e1, v0 + e2):
                        assert (p >= lo - 1e-6).all() and (p <= hi + 1e-6).all()
                    seen[prim] += 1
    return seen, depth_max


def test_bvh_builder_is_sound(built, O):
    from toyraygun_amd import capi
    for scene in (O.OracleScene.cornell_box(), O.OracleScene.cornell_lattice(4)):
        b = scene.buffers()
        nodes, tris, depth = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
        nt = scene.ntris
        assert tris.shape[0] == nt
        seen, dmax = _walk(nodes, tris, nt)
        assert (seen == 1).all() and dmax <= depth + 1
        # triangle records carry v0 / edges / mask of the right primitive
        prim = tris[:, 3:4].copy().view(np.int32)[:, 0]
        P = b["positions"].reshape(-1, 3, 3)[prim]
        assert np.array_equal(tris[:, 0:3], P[:, 0]) and np.array_equal(tris[:, 4:7], P[:, 1] - P[:, 0]) and np.array_equal(tris[:, 8:11], P[:, 2] - P[:, 0])
        assert np.array_equal(tris[:, 7:8].copy().view(np.uint32)[:, 0], b["material_ids"][prim])
    # degenerate inputs: empty scene, one triangle, many coincident triangles
    n0, t0, _ = capi.debug_build_bvh(np.zeros((0, 3), np.float32), np.zeros(0, np.uint32), np.zeros(0, np.uint32))
    assert n0.shape[0] == 1 and t0.shape[0] == 1 and t0[0, 7:8].view(np.uint32)[0] == 0
    one = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    n1, t1, _ = capi.debug_build_bvh(one, [0, 1, 2], [1])
    assert n1.shape[0] == 1 and t1.shape[0] == 1
    same = np.tile(one, (40, 1))
    n2, t2, d2 = capi.debug_build_bvh(same, np.arange(120), np.ones(40, np.uint32))
    seen, _ = _walk(n2, t2, 40)
    assert (seen == 1).all() and d2 <= 12


def _quad_leaves(nodes):
    refs = nodes[:, 12:14].copy().view(np.int32).reshape(-1)
    return sorted((~int(r)) >> 3 for r in refs if r < 0 and ((~int(r)) & 7) == 7)


def test_builder_pairs_parallelograms_into_quad_leaves(built, O):
    """Round 4: two consecutive triangles that share their first vertex and the diagonal, of one material, whose fourth corner is where
    a parallelogram has it become ONE primitive with a leaf of its own (count field 7), stored X first (the half s >= t), Y second --
    both index patterns of the reference's shapes (Scene.cpp:37-55 cube faces (a,b,c)(a,c,d); :60-92 planes / light (a,b,c)(a,d,b)).
    Everything else stays triangles: trapezoids, different materials, a shared edge that is not the diagonal, zero area, a bent
    pair, and the whole feature with TRG_BVH_QUADS=0."""
    from toyraygun_amd import capi
    a, b, c, d = (np.array(v, np.float32) for v in ([0.25, 0.5, 1], [1.25, 0.5, 1.5], [1.5, 1.75, 1.25], [0.5, 1.75, 0.75]))   # d = a + (c - b)
    far = np.array([5, 5, 5], np.float32)       # (a third primitive, so that the tree has more than the synthesised root)
    extra = [far, far + [1, 0, 0], far + [0, 1, 0]]

    def build(tris, mats=None):
        pos = np.array([v for t in tris for v in t] + extra, np.float32)
        n = len(tris) + 1
        return capi.debug_build_bvh(pos, np.arange(3 * n, dtype=np.uint32), np.array((mats or [1] * len(tris)) + [1], np.uint32))

    for pattern, x_id in (([(a, b, c), (a, c, d)], 0), ([(a, b, c), (a, d, b)], 1)):
        if x_id == 1:
            # pattern 2: (a, b', c')(a, d', b') with the diagonal b': a parallelogram a, d', b', c' -> c' = b' + ... choose b' = c, d' = b, c' = d
            pattern = [(a, c, d), (a, b, c)]
        nodes, tris, _ = build(pattern)
        q = _quad_leaves(nodes)
        assert len(q) == 1
        X, Y = tris[q[0]], tris[q[0] + 1]
        assert int(X[3:4].view(np.int32)[0]) == x_id and int(Y[3:4].view(np.int32)[0]) == 1 - x_id
        assert np.array_equal(X[0:3], a) and np.array_equal(Y[0:3], a)
        assert np.array_equal(X[8:11], Y[4:7])                             # X.e2 = Y.e1 = the diagonal
        assert np.allclose(X[4:7] + Y[8:11], X[8:11], atol=1e-6)           # X.e1 + Y.e2 = the diagonal
        seen, _ = _walk(nodes, tris, 3)
        assert (seen == 1).all()
    bent = d + np.array([0, 0, 0.01], np.float32)
    for tris, mats in (([(a, b, c), (a, c, d + np.array([0.1, 0, 0], np.float32))], None),     # a trapezoid
                       ([(a, b, c), (a, c, bent)], None),                                     # not planar
                       ([(a, b, c), (a, c, d)], [1, 2]),                                      # two materials
                       ([(a, b, c), (b, c, d)], None),                                        # no common first vertex
                       ([(a, b, c), (a, b, d)], None),                                        # the shared edge is not the diagonal
                       ([(a, b, a + 2 * (b - a)), (a, a + 2 * (b - a), a + (b - a))], None)): # zero area
        nodes, recs, _ = build(tris, mats)
        assert _quad_leaves(nodes) == []
        assert (_walk(nodes, recs, 3)[0] == 1).all()
    # a chain: (k, k+1) pair, k+2 pairs with k+1 too -- the first pair wins, k+2 stays a triangle
    e = a + (d - a) + (d - a) - (c - b) * 0      # any point: (a, d, e') below is built so that (a,c,d)(a,d,e') is a parallelogram too
    e2 = a + (d - a) + ((d - a) - (c - a))
    nodes, recs, _ = build([(a, b, c), (a, c, d), (a, d, e2)])
    assert len(_quad_leaves(nodes)) == 1 and (_walk(nodes, recs, 4)[0] == 1).all()
    os.environ["TRG_BVH_QUADS"] = "0"
    try:
        nodes, recs, _ = build([(a, b, c), (a, c, d)])
        assert _quad_leaves(nodes) == [] and (_walk(nodes, recs, 3)[0] == 1).all()
        box = O.OracleScene.cornell_box().buffers()
        assert _quad_leaves(capi.debug_build_bvh(box["positions"], box["indices"], box["material_ids"])[0]) == []
    finally:
        del os.environ["TRG_BVH_QUADS"]
    # the reference's own shapes are quads throughout
    for scene in (O.OracleScene.cornell_box(), O.OracleScene.cornell_lattice(3)):
        bf = scene.buffers()
        assert len(_quad_leaves(capi.debug_build_bvh(bf["positions"], bf["indices"], bf["material_ids"])[0])) == scene.ntris // 2


def test_builder_finds_parallelepipeds_and_keeps_the_tree_sound(built, O):
    """Round 5, BOX leaves (bvh_build.h kLeafBox): twelve consecutive triangles that are six quads bounding a parallelepiped -- Scene::addCube --
    are ONE primitive of the split rule and end up as one subtree of six quad leaves over twelve consecutive records; the builder reports
    the subtree's root, the parallelepiped's frame and which quad is which face.  tests/util.py box_zoo: rotated, sheared, mirrored, nested,
    emissive cubes are found, an almost-cube is not; with the boxes the BVH2 and its 4-wide collapse still reach every triangle exactly once."""
    from toyraygun_amd import capi
    from tests.util import box_zoo
    scene, n_boxes = box_zoo(O)
    b = scene.buffers()
    boxes = capi.debug_boxes(b["positions"], b["indices"], b["material_ids"])
    assert boxes.shape == (n_boxes, 20)
    os.environ["TRG_DEBUG_BVH_BOXES"] = "1"
    try:
        nodes, tris, depth = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
        nodes4, depth4 = capi.debug_build_bvh4(b["positions"], b["indices"], b["material_ids"])
        assert (_walk(nodes, tris, scene.ntris)[0] == 1).all()
        lat = O.OracleScene.cornell_lattice(4).buffers()
        ln, lt, _ = capi.debug_build_bvh(lat["positions"], lat["indices"], lat["material_ids"])
        assert (_walk(ln, lt, len(lat["material_ids"]))[0] == 1).all()
        assert capi.debug_boxes(lat["positions"], lat["indices"], lat["material_ids"]).shape[0] == 4 ** 3 + 2
    finally:
        del os.environ["TRG_DEBUG_BVH_BOXES"]
    prim_of = tris[:, 3:4].copy().view(np.int32)[:, 0]
    firsts = set()
    for row in boxes:
        node, first = int(row[0]), int(row[1])
        c, A = row[2:5].astype(np.float64), row[5:14].reshape(3, 3).astype(np.float64)
        prims = np.sort(prim_of[first:first + 12])
        assert np.array_equal(prims, np.arange(prims[0], prims[0] + 12)) and prims[0] % 2 == 0      # twelve consecutive triangles
        firsts.add(int(prims[0]))
        for r in range(first, first + 12):                       # every corner of the twelve sits at (+-1, +-1, +-1) of the frame
            v0, e1, e2 = tris[r, 0:3].astype(np.float64), tris[r, 4:7].astype(np.float64), tris[r, 8:11].astype(np.float64)
            for P in (v0, v0 + e1, v0 + e2):
                assert np.allclose(np.abs(A @ (P - c)), 1.0, atol=2e-5)
        for f in range(6):                                       # ... and face f = 2 k + (l_k > 0) names the quad that lies on that side
            k, side = f // 2, (1.0 if f % 2 else -1.0)
            off = int(row[14 + f])
            assert off % 2 == 0 and 0 <= off <= 10
            for r in (first + off, first + off + 1):
                v0, e1, e2 = tris[r, 0:3].astype(np.float64), tris[r, 4:7].astype(np.float64), tris[r, 8:11].astype(np.float64)
                for P in (v0, v0 + e1, v0 + e2):
                    assert abs((A @ (P - c))[k] - side) < 2e-5
        assert sorted(int(row[14 + f]) for f in range(6)) == [0, 2, 4, 6, 8, 10]
        # the subtree under `node` holds exactly the six quad leaves of these records
        leaves, st = [], [node]
        while st:
            for ref in nodes[st.pop(), 12:14].copy().view(np.int32):
                if ref >= 0:
                    st.append(int(ref))
                else:
                    leaves.append((~int(ref)) >> 3)
                    assert ((~int(ref)) & 7) == 7
        assert sorted(leaves) == [first + 2 * j for j in range(6)]
    assert firsts == {0, 12} | {36 + 12 * j for j in range(6)}    # the Cornell box's two cubes (triangles 0-11, 12-23: addCube comes first) and the zoo's six
    for name in ("TRG_BVH_BOXES", "TRG_BVH_QUADS"):               # both switches off the feature (a box is made of quads)
        os.environ[name] = "0"
        try:
            assert capi.debug_boxes(b["positions"], b["indices"], b["material_ids"]).shape[0] == 0
        finally:
            del os.environ[name]


def test_wide_bvh_is_sound(built, O):
    """The 4-wide collapse reaches every triangle exactly once, boxes enclose their subtrees, and the stack
    bound the kernel sizes its scratch from (3 * depth + 2) holds."""
    from toyraygun_amd import capi
    for scene in (O.OracleScene.cornell_box(), O.OracleScene.cornell_lattice(5), O.OracleScene()):
        b = scene.buffers()
        _, tris, _ = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
        nodes4, depth4 = capi.debug_build_bvh4(b["positions"], b["indices"], b["material_ids"])
        nt = scene.ntris
        seen = np.zeros(max(nt, 1), np.int32)
        child = nodes4[:, 24:28].copy().view(np.int32)
        dmax = 0
        stack = [(0, 1, np.full(3, -np.inf), np.full(3, np.inf))]
        visited = 0
        while stack:
            ni, d, plo, phi = stack.pop()
            visited += 1
            dmax = max(dmax, d)
            n = nodes4[ni]
            for k in range(4):
                ref = int(child[ni, k])
                if ref == -2 ** 31:
                    continue
                lo, hi = np.array([n[0 + k], n[8 + k], n[16 + k]]), np.array([n[4 + k], n[12 + k], n[20 + k]])
                assert (lo >= plo - 1e-4).all() and (hi <= phi + 1e-4).all()      # nested (up to the padding)
                if ref >= 0:
                    stack.append((ref, d + 1, lo - 1e-4, hi + 1e-4))
                else:
                    code = ~ref
                    first, count = code >> 3, (2 if (code & 7) == 7 else (code & 7) + 1)
                    for r in range(first, first + count):
                        rec = tris[r]
                        v0, e1, e2 = rec[0:3], rec[4:7], rec[8:11]
                        if nt:
                            for pnt in (v0, v0 + e1, v0 + e2):
                                assert (pnt >= lo - 1e-6).all() and (pnt <= hi + 1e-6).all()
                            seen[int(rec[3:4].view(np.int32)[0])] += 1
        assert visited == nodes4.shape[0] and dmax == depth4
        if nt:
            assert (seen == 1).all()
        assert depth4 <= 20


def test_quantised_wide_nodes_are_conservative(built, O):
    """The 64-byte quantised nodes the HBM kernels load (q4node.h): every decoded child box contains the float box it
    was made from (so no triangle the float tree reaches can be culled), is tight to within one quantisation step,
    children are unchanged, unused slots decode to an inverted box, scales are powers of two."""
    from toyraygun_amd import capi
    rng = np.random.default_rng(7)
    soup = rng.uniform(-50, 50, (300, 1, 3)).astype(np.float32) + rng.normal(0, 0.5, (300, 3, 3)).astype(np.float32)
    scenes = [O.OracleScene.cornell_box().buffers(), O.OracleScene.cornell_lattice(6).buffers(), O.OracleScene().buffers(),
              dict(positions=soup.reshape(-1, 3), indices=np.arange(900, dtype=np.uint32), material_ids=np.ones(300, np.uint32)),
              # flat (zero extent on one axis) and far from the origin
              dict(positions=np.array([[1e4, 5, 0], [1e4 + 1, 5, 0], [1e4, 5, 1], [1e4 + 2, 5, 3], [1e4 + 3, 5, 3], [1e4 + 2, 5, 4]], np.float32),
                   indices=np.arange(6, dtype=np.uint32), material_ids=np.ones(2, np.uint32))]
    for b in scenes:
        nodes4, _ = capi.debug_build_bvh4(b["positions"], b["indices"], b["material_ids"])
        q = capi.debug_build_bvh4q(b["positions"], b["indices"], b["material_ids"])
        assert q.shape == (nodes4.shape[0], 16)
        child = nodes4[:, 24:28].copy().view(np.int32)
        assert np.array_equal(q[:, 12:16].view(np.int32), child)
        origin = q[:, 0:3].copy().view(np.float32).astype(np.float64)
        scale = np.stack([q[:, 3], q[:, 10], q[:, 11]], 1).copy().view(np.float32).astype(np.float64)
        m, _e = np.frexp(scale)
        assert (m == 0.5).all()                                   # powers of two
        qd = {0: (q[:, 4], q[:, 5]), 1: (q[:, 6], q[:, 7]), 2: (q[:, 8], q[:, 9])}
        for k in range(4):
            used = child[:, k] != -2 ** 31
            for a in range(3):
                ql = ((qd[a][0] >> (8 * k)) & 255).astype(np.float64)
                qh = ((qd[a][1] >> (8 * k)) & 255).astype(np.float64)
                lo = nodes4[:, a * 8 + k].astype(np.float64)
                hi = nodes4[:, a * 8 + 4 + k].astype(np.float64)
                dlo, dhi = origin[:, a] + ql * scale[:, a], origin[:, a] + qh * scale[:, a]
                assert (dlo[used] <= lo[used]).all() and (dhi[used] >= hi[used]).all()
                assert (lo[used] - dlo[used] < scale[used, a]).all() and (dhi[used] - hi[used] < scale[used, a]).all()
                assert (ql[~used] == 255).all() and (qh[~used] == 0).all()


def test_host_builder_under_sanitizers():
    """The host BVH builder (threaded above 65,536 triangles), the 4-wide collapse and the node quantiser compiled with
    -fsanitize=address,undefined and again with -fsanitize=thread (g++, CPU only: GPU sanitizers are not available) on
    empty, tiny, degenerate-heavy and 200 k-triangle inputs."""
    import tempfile
    csrc = os.path.join(ROOT, "toyraygun_amd", "csrc")
    for san in ("address,undefined", "thread"):
        with tempfile.TemporaryDirectory() as td:
            exe = os.path.join(td, "san")
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-pthread",
                                   "-fno-omit-frame-pointer", "-I" + csrc, os.path.join(ROOT, "tests", "helpers", "bvh_sanitize.cpp"),
                                   os.path.join(csrc, "bvh_build.cpp"), "-o", exe])
            r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, TRG_BVH_THREADS="4"))
        assert r.returncode == 0, (san, r.stderr[-2000:])
        assert "n=200000" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr


def test_box_flavour_of_the_wide_tree_under_sanitizers():
    """The second 4-wide collapse (shipped build, scenes traversed from HBM): every box -- and every lone quad, dressed as a box of no thickness --
    is a leaf addressed like a record behind the records.  tests/helpers/box_tree_check.cpp builds 1 ... 9,000 cubes + a lone triangle + two lone
    quads (threaded above 65,536 triangles), and once a handful of hostile cubes, under -fsanitize=address,undefined and checks that every child code is in range and every leaf
    record is reached exactly once through exactly one leaf."""
    import tempfile
    csrc = os.path.join(ROOT, "toyraygun_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "boxcheck")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread", "-fno-omit-frame-pointer",
                               "-I" + csrc, os.path.join(ROOT, "tests", "helpers", "box_tree_check.cpp"), os.path.join(csrc, "bvh_build.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TRG_BVH_THREADS="4"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2000:])
    assert r.stdout.count("every record once: yes") == 5 and "ERROR" not in r.stderr and "runtime error" not in r.stderr
    # (cubes of no size, of no thickness, at 1e30, with a NaN or an infinite corner, of two materials, with a corner off by 1e-3: not boxes, no crash)
    assert "hostile cubes: none taken for a box" in r.stdout


def test_threaded_host_build_is_identical_to_the_single_threaded_one(built):
    """Worker threads build disjoint subtrees of the same tree: node arrays, triangle order and quantised nodes are the
    same bits for 1 and 5 threads (run in child processes: the thread count is read from the environment at build time)."""
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from toyraygun_amd import capi; from oracle import pyoracle as O;"
            "b = O.OracleScene.cornell_lattice(19).buffers();"          # 82,344 triangles: above the threading threshold
            "n, t, d = capi.debug_build_bvh(b['positions'], b['indices'], b['material_ids']);"
            "q = capi.debug_build_bvh4q(b['positions'], b['indices'], b['material_ids']);"
            "import hashlib; print(hashlib.sha256(n.tobytes() + t.tobytes() + q.tobytes()).hexdigest(), n.shape[0], d)") % ROOT
    outs = []
    for thr in ("1", "5"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, TRG_BVH_THREADS=thr))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] and len(outs[0].split()[0]) == 64


def test_halton_device_identities():
    """The two arithmetic identities behind the device Halton code, checked exhaustively in C
    (tests/helpers/halton_identities.c): fp32 digit extraction for every n < 2^22 and every prime of the table,
    and the closed form of the base-2 radical inverse on a strided sweep of all 32-bit indices."""
    import tempfile
    src = os.path.join(ROOT, "tests", "helpers", "halton_identities.c")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "hi")
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-mfma", "-ffp-contract=off", src, "-o", exe, "-lm"])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600).stdout
    assert "bad=0" in out and "bad2=0" in out and "bad3=0" in out, out


def test_bvh_is_deterministic_and_shallow_enough(built, O):
    from toyraygun_amd import capi
    b = O.OracleScene.cornell_lattice(12).buffers()   # 20,772 triangles
    a1 = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
    a2 = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
    assert np.array_equal(a1[0].view(np.uint32), a2[0].view(np.uint32)) and np.array_equal(a1[1].view(np.uint32), a2[1].view(np.uint32))
    assert a1[2] <= 40     # the kernel's LDS stack is sized from this depth (trg_capi.cpp plan_lds)


def test_shader_registry_and_engine(built):
    """The three shader names of main.cpp:24,41,56 load; others do not; compile() succeeds."""
    src = r'''
#include <stdio.h>
#include "engine/Engine.h"
#include "engine/Renderer.h"
#include "engine/Shader.h"
using namespace toyraygun;
int main() {
    Engine* e = Engine::instance();
    e->init(320, 200);
    if (e->getWidth() != 320 || e->getHeight() != 200 || e->hasQuit()) return 1;
    Shader* s = Engine::createShader();
    if (!s->load("Raytracing")) return 2;
    s->addFunction("raygen", ShaderFunctionType::RayGen);
    s->addFunction("primaryHit", ShaderFunctionType::ClosestHit);
    if (!s->compile(ShaderType::Raytrace)) return 3;
    if (s->getFunction(ShaderFunctionType::ClosestHit) != "primaryHit" || s->getFunctionNames().size() != 2) return 4;
    Shader* bad = Engine::createShader();
    if (bad->load("NoSuchShader")) return 5;
    Renderer base;
    if (base.init()) return 6;                 // base class is not a backend (Renderer.cpp:26)
    base.addShader(s);
    if (base.getShader("Raytracing") != s || base.getShader("x") != nullptr) return 7;
    e->setFrameBudget(3);
    int frames = 0;
    while (!e->hasQuit()) { e->pollEvents(); if (e->hasQuit()) break; ++frames; }
    if (frames != 3) return 8;
    printf("ok\n");
    return 0;
}
'''
    import tempfile
    from toyraygun_amd import capi
    with tempfile.TemporaryDirectory() as td:
        cpp, exe = os.path.join(td, "t.cpp"), os.path.join(td, "t")
        open(cpp, "w").write(src)
        subprocess.check_call(["g++", "-std=c++17", "-fno-exceptions", "-fno-rtti", "-I" + os.path.join(ROOT, "include"), cpp, "-o", exe,
                               "-L" + capi.LIB_DIR, "-ltoyraygun", "-ltoyraygun_hip", "-Wl,-rpath," + capi.LIB_DIR])
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under toyraygun_amd/ or include/ may import, include or link it."""
    bad = []
    for base in ("toyraygun_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if os.path.basename(dp) in ("build", "lib", "__pycache__"):
                continue
            for f in fn:
                if f.endswith((".py", ".h", ".cpp", ".hip", ".c")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    for pat in (r"import\s+oracle", r"from\s+oracle", r"pyoracle", r"liboracle", r"trg_oracle\.h", r"orc_[a-z_]+\s*\("):
                        if re.search(pat, txt):
                            bad.append((os.path.join(dp, f), pat))
    assert not bad, bad


def test_texture_load_file_decodes_png_and_ppm(built, tmp_path):
    """Texture::loadFile (reference src/engine/Texture.cpp:39-48 via stb_image; here the in-tree reader): 8-bit PNG of every
    supported colour type, written by PIL with real deflate (dynamic Huffman, adaptive filters), and binary PPM / PGM, decode
    to exactly PIL's pixels; files it cannot handle are refused, not mis-decoded."""
    from PIL import Image
    from toyraygun_amd import host
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:97, 0:131]
    smooth = np.stack([(xx * 2) % 256, (yy * 3) % 256, (xx + yy) % 256, 255 - (xx % 256)], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (97, 131, 4), dtype=np.uint8)
    for name, arr in (("smooth", smooth), ("noise", noise)):
        for mode, ch in (("L", 1), ("LA", 2), ("RGB", 3), ("RGBA", 4)):
            a = arr[..., :ch] if ch > 1 else arr[..., 0]
            p = tmp_path / ("%s_%s.png" % (name, mode))
            Image.fromarray(a, mode).save(p, compress_level=6 if name == "smooth" else 1)
            t = host.Texture(path=p)
            assert t.info() == (131, 97, ch)
            assert np.array_equal(t.pixels(), a.reshape(97, 131, ch)), (name, mode)
    p = tmp_path / "stored.png"
    Image.fromarray(noise[..., :3], "RGB").save(p, compress_level=0)     # stored deflate blocks
    assert np.array_equal(host.Texture(path=p).pixels(), noise[..., :3])
    p = tmp_path / "a.ppm"
    Image.fromarray(noise[..., :3], "RGB").save(p)
    assert np.array_equal(host.Texture(path=p).pixels(), noise[..., :3])
    p = tmp_path / "a.pgm"
    Image.fromarray(noise[..., 0], "L").save(p)
    assert np.array_equal(host.Texture(path=p).pixels()[..., 0], noise[..., 0])
    # refused: interlaced PNG, 16-bit PNG, palette PNG, truncated file, not an image
    Image.fromarray(smooth[..., :3], "RGB").convert("P").save(tmp_path / "pal.png")
    Image.fromarray(smooth[..., 0].astype(np.uint16) * 257).save(tmp_path / "d16.png")
    data = (tmp_path / "smooth_RGB.png").read_bytes()
    (tmp_path / "cut.png").write_bytes(data[: len(data) // 2])
    (tmp_path / "txt.png").write_text("not an image")
    for bad in ("pal.png", "d16.png", "cut.png", "txt.png", "missing.png"):
        with pytest.raises(IOError):
            host.Texture(path=tmp_path / bad)
    # RGBA expansion used for the upload
    assert np.array_equal(host.Texture(path=tmp_path / "smooth_L.png").rgba()[..., 1], smooth[..., 0])


def test_file_parsers_under_sanitizers(tmp_path):
    """Texture::loadFile's PNG / PPM reader and Scene::addObj read files a user hands in: a mutation fuzz of both (tests/helpers/parser_fuzz.cpp:
    bit flips, overwritten bytes, 0xFFFFFFFF length fields, truncation, splices of valid seeds) built with -fsanitize=address,undefined --
    18,000 inputs here, 120,000 when it was written: any answer is fine, a sanitizer report is not."""
    from PIL import Image
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:23, 0:31]
    smooth = np.stack([(xx * 7) % 256, (yy * 9) % 256, (xx + yy) % 256, 255 - (xx % 256)], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (23, 31, 4), dtype=np.uint8)
    Image.fromarray(smooth[..., :3], "RGB").save(tmp_path / "s_rgb.png", compress_level=6)
    Image.fromarray(noise, "RGBA").save(tmp_path / "n_rgba.png", compress_level=1)
    Image.fromarray(noise[..., 0], "L").save(tmp_path / "n_l.png", compress_level=0)
    Image.fromarray(noise[..., :3], "RGB").save(tmp_path / "a.ppm")
    Image.fromarray(noise[..., 0], "L").save(tmp_path / "a.pgm")
    (tmp_path / "m.obj").write_text("# cube\n" + "".join("v %g %g %g\n" % (x, y, z) for x in (0, 1) for y in (0, 1) for z in (0, 1)) + "vt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\nvn 1 0 0\n"
                                    "f 1 2 4 3\nf 5/1/1 6/2/1 8/3/1\nf -1//2 -2//2 -3//2 -4//2\nf 1/1 3/2 7/3 5/1\n")
    hostsrc = os.path.join(ROOT, "toyraygun_amd", "csrc", "host")
    exe = str(tmp_path / "parser_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "include"), "-I" + hostsrc, os.path.join(ROOT, "tests", "helpers", "parser_fuzz.cpp")] +
                          [os.path.join(hostsrc, f) for f in ("image_reader.cpp", "Scene.cpp", "bxmath.cpp")] + ["-o", exe])
    seeds = [str(tmp_path / f) for f in ("s_rgb.png", "n_rgba.png", "n_l.png", "a.ppm", "a.pgm", "m.obj")]
    r = subprocess.run([exe, "3000"] + seeds, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "18000 inputs parsed" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
    assert int(r.stdout.split("parsed,")[1].split()[0]) > 1000      # and the seeds themselves are files the parsers accept


def test_textured_mesh_buffers(built, tmp_path):
    """Scene::addMesh with texture coordinates + Texture, and a textured OBJ (vt): the sixth / seventh buffers line up with the
    five reference buffers; triangles added before and after stay untextured."""
    from toyraygun_amd import host
    s = host.Scene.cornell_box()
    tex = host.Texture(rgba=np.full((4, 4, 4), 200, np.uint8))
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    n = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    s.add_textured_mesh(v, n, uv, [0, 1, 2, 0, 2, 3], np.eye(4, dtype=np.float32), (1, 1, 1), 1, tex)
    s.add("cube", (0.5, 0.5, 0.5), np.eye(4, dtype=np.float32))
    obj = tmp_path / "q.obj"
    obj.write_text("v 0 0 1\nv 1 0 1\nv 0 1 1\nvt 0.25 0.5\nvt 0.75 0.5\nvt 0.25 1.5\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1\nf 1 2 3\n")
    assert s.add_obj(obj, np.eye(4, dtype=np.float32), (1, 1, 1), 1, texture=tex) == 2
    b = s.buffers()
    uvs, ids, imgs = s.texture_buffers()
    nt = b["material_ids"].shape[0]
    assert nt == 36 + 2 + 12 + 2 and uvs.shape == (3 * nt, 2) and ids.shape == (nt,)
    assert (ids[:36] == 0).all() and (ids[36:38] == 1).all() and (ids[38:50] == 0).all() and ids[50] == 1 and ids[51] == 0
    assert np.array_equal(uvs[108:114], uv[[0, 1, 2, 0, 2, 3]])
    assert np.array_equal(uvs[150:153], np.array([[0.25, 0.5], [0.75, 0.5], [0.25, 1.5]], np.float32))
    assert len(imgs) == 1 and imgs[0].shape == (4, 4, 4)


def test_device_group_worker_hand_off(tmp_path):
    """The persistent host threads of trg_group_* (trg_workers.h): hand-off, completion, per-rank results, thread affinity, 3,000 rounds
    on 1 / 2 / 3 / 8 workers -- plain and under ThreadSanitizer."""
    src = os.path.join(ROOT, "tests", "helpers", "workers_test.cpp")
    for name, flags in (("plain", ["-O2"]), ("tsan", ["-O1", "-g", "-fsanitize=thread"])):
        exe = str(tmp_path / ("workers_" + name))
        subprocess.check_call(["g++", "-std=c++17", "-pthread"] + flags + [src, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "workers ok" in r.stdout, (name, r.stdout[-2000:], r.stderr[-2000:])


def test_one_band_rule(built):
    """dist.band_rows (one process per GPU) and trg_band_rows (device groups behind the C ABI) are the same arithmetic: bands of
    ceil(h / n) rows, the tail shorter or empty."""
    from toyraygun_amd import capi
    from toyraygun_amd.dist import band_rows
    capi.load()
    for h in (1, 2, 5, 17, 30, 34, 64, 135, 1079, 1080, 2160, 4321):
        for n in (1, 2, 3, 4, 5, 7, 8, 16, 64):
            rows = [band_rows(h, n, r) for r in range(n)]
            assert rows == [capi.band_rows(h, n, r) for r in range(n)], (h, n)
            assert rows[0][0] == 0 and sum(k for _, k in rows) == h
            B = -(-h // n)
            assert all(k in (B, h - r0) or k == 0 for r0, k in rows)


def test_bench_runs_plainly_with_several_gpus(built, monkeypatch, capsys, tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher must not die in argument handling (the driver's scaling run invokes it like
    that): it takes the single-process device group (trg_group_*).  Rehearsed with a stand-in for the group so that no GPU is needed:
    the JSON line, its sharding description, per-rank ray counts and the separated gather time are produced; too few devices is a
    clear error with a non-zero exit."""
    import importlib
    import json
    import torch
    from toyraygun_amd import capi
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    calls = []

    class FakeGroup:
        def __init__(self, devices, w, h):
            self.n, self.w, self.h = len(devices), w, h
            calls.append(("create", list(devices), w, h))
        def load_scene(self, *a): calls.append(("scene", len(a[4])))
        def set_uniforms(self, u): pass
        def set_pixel_offsets_seed(self, seed=0): pass
        def set_option(self, o, v): calls.append(("opt", o, v))
        exchange, exchange_note = capi.EXCHANGE_RCCL, ""
        def rccl_info(self): return {"version": 22203, "ranks": [self.n] * self.n}
        def render(self, f0, spp, b, gather=capi.GATHER_ALL, root=0):
            calls.append(("render", gather))
            if self.timing: self.timed += 1
        def sync(self): pass
        def stats(self):
            st = capi.Stats()
            st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits = 1000, 500, 700, 600
            st.node_fetches, st.tri_tests, st.wave_node_iters, st.scene_in_lds, st.scene_bytes, st.last_render_ms = 9000, 3000, 100, 1, 9900, 0.5
            st.total_render_ms, st.renders, st.last_frame_split = 1.5, 3, 1
            return st
        def rank_stats(self, r): return self.stats()
        def reset_stats(self): pass
        def rank_rays(self): return [1100, 1100]
        def set_pipeline(self, depth): calls.append(("pipeline", depth))
        def set_bands(self, mode): calls.append(("bands", mode))
        def fence_wait(self, slot): pass
        def fence_record(self, slot): pass
        timing, timed = False, 0
        def time_launches(self, on=True): self.timing = bool(on)
        def launch_ms(self, rank):
            k, self.timed = (self.timed if rank == self.n - 1 else None), (0 if rank == self.n - 1 else self.timed)
            return [0.25] * (k if k is not None else self.timed)
        def close(self): calls.append(("close",))

    monkeypatch.setattr(capi, "Group", FakeGroup)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TRG_BENCH_DETAIL", str(tmp_path / "detail.json"))
    bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    assert len(line) < 4096                                           # the driver keeps an 8 KB tail: the whole line must sit in it
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "strong" and out["unit"] == "Mrays/s"
    # the driver's record keeps the SCALARS directly under config / roofline: nothing a reader needs may hide in a nested object
    assert all(not isinstance(v, (dict, list)) for v in out["config"].values()) and all(not isinstance(v, (dict, list)) for v in out["roofline"].values())
    cfgo = out["config"]
    assert "1 process" in cfgo["sharding"] and "interleaved" in cfgo["sharding"] and cfgo["exchange"] == "rccl" and "root gather" in cfgo["exchange_mode"]
    assert cfgo["rccl_version"] == 22203 and cfgo["rccl_ranks"] == 2                     # "did RCCL see N ranks" is answerable from the line
    assert cfgo["gather_root_ms_per_step"] >= 0 and cfgo["gather_all_ms_per_step"] >= 0 and "bound" in out["roofline"]   # (the stand-in renders in no time: its rates mean nothing)
    assert ("create", [0, 1], 1920, 1080) in calls and ("close",) in calls
    assert ("pipeline", 4) in calls and "4 frames in flight" in cfgo["pipeline"]
    assert ("bands", capi.BANDS_INTERLEAVED) in calls
    detail = json.load(open(tmp_path / "detail.json"))
    assert detail["line"]["value"] == out["value"] and detail["per_rank_rays_per_step"] == [1100, 1100] and detail["rccl"]["ranks"] == [2, 2]
    bands = detail["bands"]
    assert len(bands["kernel_alone_ms"]["per_rank"]) == 2 and bands["kernel_ms_in_pipeline"]["max_over_mean"] == 1.0 and bands["rays"]["per_rank"] == [1100.0, 1100.0]
    # the TIMED exchange is north_star's gather to one root: warm-up 1 + timed 3; the all-gather beside it: 3; without an exchange: counters 1 + alone 3 + priming 4 + timed 3
    n_of = lambda mode: sum(1 for c in calls if c == ("render", mode))
    assert n_of(capi.GATHER_ROOT) == 1 + 3 and n_of(capi.GATHER_ALL) == 3 and n_of(capi.GATHER_NONE) == 1 + 3 + 4 + 3
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--no-cpu-baseline"])
    assert "only 1 GPU" in str(e.value)
    monkeypatch.setenv("TRG_BENCH_DEVICES", "0,0")   # the one-GPU rehearsal: an explicit device list (contexts sharing a device)
    bench.main(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    out = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert out["config"]["devices"] == "0,0" and out["config"]["shared_device"] is True and ("create", [0, 0], 1920, 1080) in calls
    monkeypatch.setenv("TRG_BENCH_DEVICES", "0")
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "2", "--no-cpu-baseline"])
    monkeypatch.delenv("TRG_BENCH_DEVICES")
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("MASTER_PORT", "1"); monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:   # under a launcher the world size must match
        bench.main(["--gpus", "2", "--no-cpu-baseline"])
    assert "WORLD_SIZE=1" in str(e.value)


def test_texture_images_follow_the_native_registration_order(built, tmp_path):
    """Scene::textureID registers a texture when a face with texture coordinates first USES it.  An OBJ without `vt`
    faces (or an unreadable file) given a texture registers nothing, so the image list handed to trg_load_textures must come from
    the native list, not from the order Python saw the textures in -- otherwise texture id 1 would upload the wrong picture."""
    from toyraygun_amd import host
    s = host.Scene.cornell_box()
    t1 = host.Texture(rgba=np.full((2, 2, 4), 11, np.uint8))
    t2 = host.Texture(rgba=np.full((3, 3, 4), 22, np.uint8))
    plain = tmp_path / "plain.obj"
    plain.write_text("v 0 0 1\nv 1 0 1\nv 0 1 1\nf 1 2 3\n")                    # no vt: t1 is never registered
    assert s.add_obj(plain, np.eye(4, dtype=np.float32), (1, 1, 1), 1, texture=t1) == 1
    assert s.add_obj(tmp_path / "missing.obj", np.eye(4, dtype=np.float32), (1, 1, 1), 1, texture=t1) == -1
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]], np.float32)
    n = np.tile(np.array([[0, 0, 1]], np.float32), (3, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1]], np.float32)
    s.add_textured_mesh(v, n, uv, [0, 1, 2], np.eye(4, dtype=np.float32), (1, 1, 1), 1, t2)   # native id 1 = t2
    uvs, ids, imgs = s.texture_buffers()
    assert ids[-1] == 1 and (ids[:-1] == 0).all()
    assert len(imgs) == 1 and imgs[0].shape == (3, 3, 4) and (imgs[0] == 22).all()
    s.add_textured_mesh(v, n, uv, [0, 1, 2], np.eye(4, dtype=np.float32), (1, 1, 1), 1, t1)   # now t1 becomes id 2
    _, ids, imgs = s.texture_buffers()
    assert ids[-1] == 2 and [im.shape[0] for im in imgs] == [3, 2]


def test_png_header_cannot_size_the_decoder(built, tmp_path):
    """a tiny file whose IHDR claims 32768 x 32768 x 4 must be refused without reserving 4 GiB (this library is built
    without exceptions: a failed allocation would end the process), and a deflate stream longer than the header implies stops
    at the implied size."""
    import struct
    import zlib
    from toyraygun_amd import host

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    def png(w, h, raw):
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")
    (tmp_path / "huge.png").write_bytes(png(32768, 32768, b"\0" * 64))               # 4 GiB claimed, 64 bytes delivered
    (tmp_path / "bomb.png").write_bytes(png(4, 4, b"\0" * (64 << 20)))               # 68 bytes implied, 64 MiB delivered
    (tmp_path / "ok.png").write_bytes(png(4, 4, (b"\0" + b"\x10\x20\x30\xff" * 4) * 4))
    for bad in ("huge.png",):
        with pytest.raises(IOError):
            host.Texture(path=tmp_path / bad)
    with pytest.raises(IOError):
        host.Texture(path=tmp_path / "bomb.png")
    t = host.Texture(path=tmp_path / "ok.png")
    assert t.info() == (4, 4, 4) and (t.pixels()[..., 0] == 0x10).all()


def test_committed_counters_belong_to_these_kernel_sources(built):
    """Evidence hygiene (round-2 verdict, item 8): the profiler counters bench.py imports into its roofline object carry a hash of the
    kernel sources they were measured on; the committed files must match THIS tree (re-run scripts/profile_round.sh after touching
    a kernel), and a file with another hash is refused and reported as stale rather than used."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from toyraygun_amd.srchash import kernel_source_hash
    here = kernel_source_hash()
    for cfg in ("c2", "c3", "c4"):
        rec, path, stale = bench.imported_counters(cfg)
        assert rec is not None and stale is None, (cfg, stale and stale.get("reason"))
        assert rec["kernel_source_hash"] == here and rec["valu_insts_per_launch"] > 0 and path.startswith("profiles/")
    # a counters file of another build is not used
    rounds = bench.PROFILE_ROUNDS
    try:
        bench.PROFILE_ROUNDS = ("r02",)          # round 2's files carry no hash at all
        rec, path, stale = bench.imported_counters("c2")
        assert rec is None and stale is not None and "changed" in stale["reason"]
    finally:
        bench.PROFILE_ROUNDS = rounds


def test_every_tile_order_is_a_bijection(built):
    """TRG_OPT_TILE_ORDER (host replica of the kernels' slot -> tile map, trg_kernels.h tile_of_slot): for every order the valid slots of
    a launch cover every tile exactly once; the XCD-aware orders pad by less than one column / row of tiles per region, give the slots
    with equal slot % 8 ONE rectangle of the screen, and walk it away from the image centre."""
    import ctypes as C
    from toyraygun_amd import capi
    L = capi.load()
    n, bx, by = C.c_uint32(), C.c_uint32(), C.c_uint32()
    for tx, ty in ((120, 68), (1, 1), (7, 3), (13, 1), (2, 9), (240, 135), (5, 5), (64, 2)):
        for order in (0, 1, 2, 4, 8, 17, 18, 20, 24, 32):
            assert L.trg_debug_tile_of_slot(tx, ty, order, 0, C.byref(n), None, None) >= 0
            slots = n.value
            seen, per_xcd = set(), {}
            for s_ in range(slots):
                rc = L.trg_debug_tile_of_slot(tx, ty, order, s_, None, C.byref(bx), C.byref(by))
                assert rc in (0, 1)
                if rc:
                    assert bx.value < tx and by.value < ty and (bx.value, by.value) not in seen
                    seen.add((bx.value, by.value))
                    per_xcd.setdefault(s_ % 8, []).append((bx.value, by.value))
            assert len(seen) == tx * ty, (tx, ty, order)
            assert L.trg_debug_tile_of_slot(tx, ty, order, slots, None, None, None) == 0
            if order == 0:
                assert slots == tx * ty
            elif order == 32:
                # 8 x 8-tile super-blocks, one 2 x 4-tile sub-block per XCD: the 8 tiles an XCD gets from a super-block form one 2 x 4 rectangle
                assert slots == ((tx + 7) // 8) * ((ty + 7) // 8) * 64
                for x, tiles in per_xcd.items():
                    groups = {}
                    for (a, b_) in tiles:
                        groups.setdefault((a // 8, b_ // 8), []).append((a, b_))
                    for g_ in groups.values():
                        xs, ys = [t[0] for t in g_], [t[1] for t in g_]
                        assert max(xs) - min(xs) <= 1 and max(ys) - min(ys) <= 3
                if tx % 8 == 0 and ty >= 16:
                    counts = [len(v) for v in per_xcd.values()]
                    assert max(counts) - min(counts) <= tx // 8 * 8 // 2 + 8           # the partial bottom row of super-blocks is shared out by the rotation
            elif order >= 16:
                # stripes of S tile rows dealt round robin: XCD x owns exactly the rows whose group number is x mod 8, all columns of them
                S = order - 16
                assert slots <= (ty + 8 * S) * tx
                for x, tiles in per_xcd.items():
                    assert all((t[1] // S) % 8 == x for t in tiles)
                    assert len(tiles) == tx * sum(1 for r in range(ty) if (r // S) % 8 == x)
                    assert tiles[0][0] == (tx - 1) // 2                 # starts at the centre column
            else:
                assert slots <= tx * ty + 8 * (tx + ty + 1)          # padding: at most a column + a row of tiles per region
                for tiles in per_xcd.values():                        # one rectangle per XCD, filled completely
                    xs, ys = [t[0] for t in tiles], [t[1] for t in tiles]
                    assert (max(xs) - min(xs) + 1) * (max(ys) - min(ys) + 1) == len(tiles)
                if order == 2 and tx >= 4 and ty >= 4:
                    left = [t[0] for t in per_xcd[0]]                 # strip left of the centre: starts at the centre column, ends at the edge
                    assert left[0] == tx // 2 - 1 and left[-1] == 0
    assert L.trg_debug_tile_of_slot(4, 4, 3, 0, None, None, None) < 0


def test_plane_records_are_the_triangles(built, O):
    """The PLANE form of the triangle records (round 4: the shipped build's triangle test on LDS-resident scenes, trg_device.h
    tri_test_planes) against the Moeller-Trumbore records of the same build, in float64: the first plane is the unit supporting
    plane, the second and third give the weights of vertex 1 and vertex 2 (0 / 1 / 0 and 0 / 0 / 1 at the corners), the u16 carries
    the original index and the two mask bits; the test evaluated in float64 on random rays picks the hits of an independent
    Moeller-Trumbore; a degenerate triangle can never be hit."""
    from toyraygun_amd import capi
    box = O.OracleScene.cornell_box()
    rng = np.random.default_rng(5)
    soup = O.OracleScene()
    eye = np.eye(4, dtype=np.float32)
    for k in range(60):
        tri = (rng.uniform(-1, 1, 3) + rng.normal(0, 0.2, (3, 3))).astype(np.float32)
        soup.add_geometry(tri, [0, 1, 2], eye, (0.5, 0.5, 0.5), 1 + k % 2)
    soup.add_geometry(np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2]], np.float32), [0, 1, 2], eye, (0.5, 0.5, 0.5), 1)   # zero area
    for scene in (box, soup):
        b = scene.buffers()
        nodes, tris, _ = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
        planes, meta, ctr = capi.debug_plane_records(b["positions"], b["indices"], b["material_ids"])
        # QUAD leaves (bvh_build.h, count field 7): record X carries the planes of the parallelogram X.e1 x Y.e2, Y (the next record) its own
        refs = nodes[:, 12:14].copy().view(np.int32).reshape(-1)
        quad_x = np.zeros(tris.shape[0], bool)
        quad_x[[(~int(r)) >> 3 for r in refs if r < 0 and ((~int(r)) & 7) == 7]] = True
        assert quad_x.sum() == (scene.ntris // 2 if scene is box else 0)       # the box is quads throughout (Scene.cpp:24-92), the soup has none
        pos_all = b["positions"].reshape(-1, 3)
        assert np.allclose(ctr, 0.5 * (pos_all.min(0) + pos_all.max(0)), atol=1e-6)
        assert planes.shape == (tris.shape[0], 12) and meta.shape == (tris.shape[0],)
        prim, mask = tris[:, 3].view(np.uint32), tris[:, 7].view(np.uint32)
        assert np.array_equal(meta, ((prim << 2) | (mask & 3)).astype(np.uint16))
        assert sorted(prim.tolist()) == list(range(scene.ntris))
        v0, e1, e2 = tris[:, 0:3].astype(np.float64) - ctr.astype(np.float64), tris[:, 4:7].astype(np.float64), tris[:, 8:11].astype(np.float64)   # (the planes are relative to the centre)
        mt_e2 = e2.copy()                                        # (the triangles themselves, for the Moeller-Trumbore side)
        qx = np.nonzero(quad_x)[0]
        assert np.allclose(v0[qx], v0[qx + 1]) and np.allclose(e2[qx], e1[qx + 1], atol=1e-6)      # X = (p0, p1, diagonal), Y = (p0, diagonal, p3)
        assert np.allclose(e1[qx] + e2[qx + 1], e2[qx], atol=1e-5) and np.array_equal(mask[qx], mask[qx + 1])   # a parallelogram of one material
        e2[qx] = e2[qx + 1]                                      # the axes of a quad's planes: X.e1 and Y.e2
        P = planes.astype(np.float64)
        area = np.linalg.norm(np.cross(e1, e2), axis=1)
        ok = area > 0
        assert (~ok).sum() == (1 if scene is soup else 0)
        assert np.array_equal(planes[~ok], np.tile(np.array([0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1], np.float32), ((~ok).sum(), 1)))
        n, d0, n1, d1, n2, d2 = P[ok, 0:3], P[ok, 3], P[ok, 4:7], P[ok, 7], P[ok, 8:11], P[ok, 11]
        a, b1, b2 = v0[ok], v0[ok] + e1[ok], v0[ok] + e2[ok]
        scale = 1.0 + np.abs(a).max()
        assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
        for corner, (wu, wv) in ((a, (0, 0)), (b1, (1, 0)), (b2, (0, 1))):
            assert np.allclose((n * corner).sum(1), d0, atol=3e-6 * scale)
            size = np.maximum(np.linalg.norm(n1, axis=1), np.linalg.norm(n2, axis=1)) * scale
            assert np.all(np.abs((n1 * corner).sum(1) + d1 - wu) <= 4e-7 * size + 1e-6)
            assert np.all(np.abs((n2 * corner).sum(1) + d2 - wv) <= 4e-7 * size + 1e-6)
        # the test itself, in float64, against Moeller-Trumbore on the same triangles
        o = rng.uniform(-1.5, 1.5, (400, 3)); d = rng.normal(size=(400, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        for k in range(P.shape[0]):
            den = P[k, 0:3] @ d.T
            with np.errstate(divide="ignore", invalid="ignore"):
                t = -((P[k, 0:3] @ o.T) - P[k, 3]) / den
                X = o + t[:, None] * d
                u, v = X @ P[k, 4:7] + P[k, 7], X @ P[k, 8:11] + P[k, 11]
                def moller(j):
                    pv = np.cross(d, mt_e2[j]); det = pv @ e1[j]; tv = o - v0[j]
                    mu = (tv * pv).sum(1) / det; q = np.cross(tv, e1[j]); mv = (d * q).sum(1) / det; mt = (q @ mt_e2[j]) / det
                    return mu, mv, mt, (mu >= 0) & (mv >= 0) & (mu + mv <= 1) & (mt >= 0), np.minimum.reduce([np.abs(mu), np.abs(mv), np.abs(1 - mu - mv), np.abs(mt)]) < 1e-4
                if quad_x[k]:
                    # ONE test for the two triangles: inside the unit square of (s, t); s >= t is X with weights (s - t, t), t > s is Y with (s, t - s)
                    second = u < v
                    hit = (u >= 0) & (v >= 0) & (u <= 1) & (v <= 1) & (t >= 0)
                    u, v = np.where(second, u, u - v), np.where(second, v - u, v)
                    mx, my = moller(k), moller(k + 1)
                    mu, mv, mt = (np.where(second, my[i], mx[i]) for i in range(3))
                    mhit = np.where(second, my[3], mx[3])
                    edge = mx[4] | my[4]      # fp32 planes: undecided this close to an edge (the diagonal included: either triangle may claim it)
                    assert not (mx[3] & my[3] & ~edge).any()
                    assert np.array_equal((mx[3] | my[3])[~edge], hit[~edge]), k
                else:
                    hit = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= 0)
                    mu, mv, mt, mhit, edge = moller(k)
            assert np.array_equal(hit[~edge], mhit[~edge]), k
            both = hit & mhit & ~edge
            assert np.allclose(t[both], mt[both], rtol=1e-4, atol=1e-5) and np.allclose(u[both], mu[both], atol=1e-4) and np.allclose(v[both], mv[both], atol=1e-4)
            if not ok[k]:
                assert not hit.any()


def test_leaf_records_carry_each_triangles_geometry_and_attributes(built, O):
    """The 128-byte leaf records of HBM-resident scenes (round 3): every triangle exactly once, in the leaf order of the BVH; rows 0-2 the
    geometry record (v0 | original index, e1 | material id, e2), then THAT triangle's normals and colours from the reference's
    per-corner buffers -- checked on the lattice scene (distinct colours per cube) and on a soup with per-vertex attributes."""
    import ctypes as C
    from toyraygun_amd import capi
    L = capi.load()
    rng = np.random.default_rng(5)
    scenes = [O.OracleScene.cornell_lattice(3).buffers()]
    nt = 200
    soup = dict(positions=rng.uniform(-1, 1, (3 * nt, 3)).astype(np.float32), normals=rng.normal(size=(3 * nt, 3)).astype(np.float32),
                colors=rng.uniform(0, 1, (3 * nt, 3)).astype(np.float32), indices=rng.permutation(3 * nt).astype(np.uint32),
                material_ids=rng.choice([1, 2, 3], nt).astype(np.uint32))
    scenes.append(soup)
    for b in scenes:
        pos, nrm, col = (np.ascontiguousarray(b[k], np.float32) for k in ("positions", "normals", "colors"))
        idx, mat = np.ascontiguousarray(b["indices"], np.uint32), np.ascontiguousarray(b["material_ids"], np.uint32)
        n = C.c_uint32()
        args = (pos.ctypes.data, nrm.ctypes.data, col.ctypes.data, idx.ctypes.data, mat.ctypes.data, pos.shape[0], mat.shape[0])
        assert L.trg_debug_leaf_records(*args, None, 0, C.byref(n)) == capi.OK and n.value == mat.shape[0]
        rec = np.zeros((n.value, 32), np.float32)
        assert L.trg_debug_leaf_records(*args, rec.ctypes.data, n.value, C.byref(n)) == capi.OK
        prim = rec[:, 3].view(np.uint32)
        assert sorted(prim.tolist()) == list(range(mat.shape[0]))                       # a permutation: every triangle once
        v = pos[idx.reshape(-1, 3)[prim]]                                                # [n, 3 corners, 3]
        assert np.array_equal(rec[:, 0:3], v[:, 0]) and np.array_equal(rec[:, 4:7], v[:, 1] - v[:, 0]) and np.array_equal(rec[:, 8:11], v[:, 2] - v[:, 0])
        assert np.array_equal(rec[:, 7].view(np.uint32), mat[prim])                      # the mask IS the material id
        assert np.array_equal(rec[:, 12:21], nrm.reshape(-1, 9)[prim]) and np.array_equal(rec[:, 21:30], col.reshape(-1, 9)[prim])
        _, tris, _ = capi.debug_build_bvh(pos, idx, mat)                                 # same order as the 48-byte records of the LDS path
        assert np.array_equal(rec[:, :12].view(np.uint32), tris.view(np.uint32))
