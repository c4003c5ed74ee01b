"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

Test infrastructure only: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the toyraygun_amd package (tests/test_no_oracle_in_product.py
enforces that).  PARITY STATUS: "parity unpinned" (see oracle/trg_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Uniforms(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("frameIndex", C.c_uint32), ("_pad", C.c_uint32),
        ("cam_pos", C.c_float * 4),
        ("inv_view_proj", C.c_float * 16),
        ("light_pos", C.c_float * 4), ("light_forward", C.c_float * 4), ("light_right", C.c_float * 4),
        ("light_up", C.c_float * 4), ("light_color", C.c_float * 4),
    ]


assert C.sizeof(Uniforms) == 176

RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("mask", "<u4"), ("direction", "<f4", 3),
                      ("maxDistance", "<f4"), ("color", "<f4", 4)])
ISECT_DTYPE = np.dtype([("distance", "<f4"), ("primitiveIndex", "<i4"), ("coordinates", "<f4", 2)])
assert RAY_DTYPE.itemsize == 48 and ISECT_DTYPE.itemsize == 16


class Scene(C.Structure):
    _fields_ = [
        ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("colors", C.POINTER(C.c_float)),
        ("indices", C.POINTER(C.c_uint32)), ("material_ids", C.POINTER(C.c_uint32)),
        ("nverts", C.c_uint32), ("ntris", C.c_uint32), ("cap_verts", C.c_uint32), ("cap_tris", C.c_uint32),
        ("accel", C.c_void_p),
        ("uvs", C.POINTER(C.c_float)), ("tex_ids", C.POINTER(C.c_uint32)), ("tex_table", C.POINTER(C.c_uint32)),
        ("texels", C.POINTER(C.c_uint32)), ("ntextures", C.c_uint32),
        ("xnodes", C.POINTER(C.c_float)), ("xtris", C.POINTER(C.c_float)), ("xn_nodes", C.c_uint32), ("xn_recs", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("tri_tests", C.c_uint64), ("node_visits", C.c_uint64)]

    @property
    def rays(self):
        return self.primary_rays + self.bounce_rays + self.shadow_rays

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


TRIG_LIBM, TRIG_PORTABLE = 0, 1
SEED_OFFSETS = 0x5EED0001
SEED_LATTICE = 0x5EED0002
EYE = (0.0, 1.0, 3.38)   # src/main.cpp:85
AT = (0.0, 1.0, -1.0)    # src/main.cpp:86


def build(force=False, tuned=False):
    """liboracle.so (the checker: -O2, no contraction) or, tuned=True, liboracle_tuned.so -- the same source at -O3 -march=x86-64-v3 with
    contraction, what bench.py's `tuned` CPU-baseline leg times (oracle/Makefile)."""
    name = "liboracle_tuned.so" if tuned else "liboracle.so"
    so = os.path.join(_HERE, name)
    src = [os.path.join(_HERE, f) for f in ("trg_oracle.c", "trg_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return so


_LIB_TUNED = None


def tuned_lib():
    """The tuned build, bound for what the CPU-baseline leg needs: orc_render on a scene of the default library (same source, same
    structures) and orc_scene_set_bvh."""
    global _LIB_TUNED
    if _LIB_TUNED is None:
        lib()
        L = C.CDLL(build(tuned=True))
        L.orc_render.argtypes = [C.POINTER(Scene), C.POINTER(Uniforms), C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Stats), C.c_int]
        L.orc_render.restype = C.c_int
        L.orc_intersect_nearest.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats)]
        L.orc_intersect_any.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats)]
        L.orc_num_threads.restype = C.c_int
        _LIB_TUNED = L
    return _LIB_TUNED


def set_bvh(scene, nodes=None, tris=None):
    """orc_scene_set_bvh: walk this BVH2 (toyraygun_amd.capi.debug_build_bvh's arrays: nodes [n,16], records [m,12]) instead of brute force /
    the oracle's own tree; None removes it."""
    L = lib()
    if nodes is None:
        L.orc_scene_set_bvh(scene.p, None, 0, None, 0)
        return
    nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1, 16)
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 12)
    L.orc_scene_set_bvh(scene.p, _fp(nodes), nodes.shape[0], _fp(tris), tris.shape[0])


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        L.orc_scene_new.restype = C.POINTER(Scene)
        L.orc_scene_free.argtypes = [C.POINTER(Scene)]
        L.orc_scene_cornell_box.argtypes = [C.POINTER(Scene)]
        L.orc_scene_cornell_lattice.argtypes = [C.POINTER(Scene), C.c_int, C.c_uint32]
        for n in ("orc_scene_add_cube", "orc_scene_add_plane", "orc_scene_add_area_light"):
            getattr(L, n).argtypes = [C.POINTER(Scene), fp, fp]
        L.orc_scene_add_geometry.argtypes = [C.POINTER(Scene), fp, C.POINTER(C.c_uint32), C.c_int, fp, fp, C.c_uint32]
        L.orc_make_uniforms.argtypes = [C.POINTER(Uniforms), C.c_uint32, C.c_uint32, C.c_uint32, fp, fp]
        L.orc_pcg_hash32.argtypes = [C.c_uint32]
        L.orc_pcg_hash32.restype = C.c_uint32
        L.orc_pixel_offsets.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_halton.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_halton.restype = C.c_float
        L.orc_halton_prime.argtypes = [C.c_uint32]
        L.orc_halton_prime.restype = C.c_uint32
        L.orc_sample_cosine_hemisphere.argtypes = [fp, fp]
        L.orc_align_hemisphere.argtypes = [fp, fp, fp]
        L.orc_sample_area_light.argtypes = [C.POINTER(Uniforms), fp, fp, fp, fp, fp, fp]
        L.orc_sincos_portable.argtypes = [C.c_float, fp, fp]
        L.orc_mtx_srt.argtypes = [fp] + [C.c_float] * 9
        L.orc_mtx_look_at_rh.argtypes = [fp, fp, fp, fp]
        L.orc_mtx_proj_rh.argtypes = [fp] + [C.c_float] * 4
        L.orc_mtx_mul.argtypes = [fp, fp, fp]
        L.orc_mtx_inverse.argtypes = [fp, fp]
        L.orc_vec4_mul_mtx.argtypes = [fp, fp, fp]
        L.orc_raygen.argtypes = [C.POINTER(Uniforms), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_intersect_nearest.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats)]
        L.orc_intersect_any.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats)]
        L.orc_intersect_nearest_brute.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_intersect_any_brute.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_nearest_f64.argtypes = [C.POINTER(Scene), C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_nearest_f64.restype = C.c_int
        L.orc_render.argtypes = [C.POINTER(Scene), C.POINTER(Uniforms), C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Stats), C.c_int]
        L.orc_render.restype = C.c_int
        L.orc_postprocess.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_set_trig_mode.argtypes = [C.c_int]
        L.orc_scene_set_bvh.argtypes = [C.POINTER(Scene), fp, C.c_uint32, fp, C.c_uint32]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleScene:
    """Owns an orc_scene; exposes the five Scene.h buffers as numpy copies."""

    def __init__(self):
        self.L = lib()
        self.p = self.L.orc_scene_new()

    def __del__(self):
        try:
            self.L.orc_scene_free(self.p)
        except Exception:
            pass

    @classmethod
    def cornell_box(cls):
        s = cls()
        s.L.orc_scene_cornell_box(s.p)
        return s

    @classmethod
    def cornell_lattice(cls, n, seed=SEED_LATTICE):
        s = cls()
        s.L.orc_scene_cornell_lattice(s.p, int(n), seed)
        return s

    def add(self, kind, color, mtx):
        c = np.asarray(color, np.float32).copy()
        m = np.asarray(mtx, np.float32).reshape(16).copy()
        getattr(self.L, {"cube": "orc_scene_add_cube", "plane": "orc_scene_add_plane",
                         "light": "orc_scene_add_area_light"}[kind])(self.p, _fp(c), _fp(m))

    def add_geometry(self, verts, tri_idx, mtx, color, material_id):
        v = np.ascontiguousarray(verts, np.float32)
        t = np.ascontiguousarray(tri_idx, np.uint32)
        m = np.asarray(mtx, np.float32).reshape(16).copy()
        c = np.asarray(color, np.float32).copy()
        self.L.orc_scene_add_geometry(self.p, _fp(v), t.ctypes.data_as(C.POINTER(C.c_uint32)), t.size // 3,
                                      _fp(m), _fp(c), material_id)

    def add_raw(self, positions, normals, colors, material_ids):
        """Append unindexed triangles exactly as given (the five Scene.h vectors are public): per-corner positions,
        normals and colours [3*n, 3], material ids [n]."""
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        c = np.ascontiguousarray(colors, np.float32).reshape(-1, 3)
        m = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
        assert p.shape[0] == 3 * m.shape[0] == n.shape[0] == c.shape[0]
        self.L.orc_scene_add_raw.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.L.orc_scene_add_raw(self.p, p.ctypes.data, n.ctypes.data, c.ctypes.data, m.ctypes.data, m.shape[0])

    def set_textures(self, uvs, texture_ids, images):
        """Albedo textures (project definition, see trg_oracle.h): uvs [3*ntris, 2], ids [ntris], images list of [h, w, 4] uint8."""
        uv = np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(texture_ids, np.uint32).reshape(-1)
        assert uv.shape[0] == 3 * self.ntris and ids.shape[0] == self.ntris
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        ptrs = (C.c_void_p * max(len(imgs), 1))(*[im.ctypes.data for im in imgs])
        ws = np.array([im.shape[1] for im in imgs], np.uint32)
        hs = np.array([im.shape[0] for im in imgs], np.uint32)
        self.L.orc_scene_set_textures.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_uint32]
        self.L.orc_scene_set_textures(self.p, uv.ctypes.data, ids.ctypes.data, ptrs, ws.ctypes.data, hs.ctypes.data, len(imgs))

    @property
    def ntris(self):
        return int(self.p.contents.ntris)

    @property
    def nverts(self):
        return int(self.p.contents.nverts)

    def buffers(self):
        s = self.p.contents
        nv, nt = s.nverts, s.ntris
        if nv == 0:
            return dict(positions=np.zeros((0, 3), np.float32), normals=np.zeros((0, 3), np.float32),
                        colors=np.zeros((0, 3), np.float32), indices=np.zeros(0, np.uint32), material_ids=np.zeros(0, np.uint32))
        return dict(
            positions=np.ctypeslib.as_array(s.positions, (nv, 3)).copy(),
            normals=np.ctypeslib.as_array(s.normals, (nv, 3)).copy(),
            colors=np.ctypeslib.as_array(s.colors, (nv, 3)).copy(),
            indices=np.ctypeslib.as_array(s.indices, (nv,)).copy(),
            material_ids=np.ctypeslib.as_array(s.material_ids, (nt,)).copy(),
        )


def make_uniforms(w, h, frame_index=0, eye=EYE, at=AT):
    u = Uniforms()
    e = np.asarray(eye, np.float32)
    a = np.asarray(at, np.float32)
    lib().orc_make_uniforms(C.byref(u), w, h, frame_index, _fp(e), _fp(a))
    return u


def uniforms_bytes(u):
    return bytes(memoryview(u))


def pixel_offsets(w, h, seed=SEED_OFFSETS):
    out = np.empty(w * h, np.uint32)
    lib().orc_pixel_offsets(seed, w, h, out.ctypes.data)
    return out


def halton(i, d):
    return float(lib().orc_halton(int(i) & 0xFFFFFFFF, int(d)))


def render(scene, w, h, spp, bounces, frame_begin=0, row0=0, rows=None, accum=None, offsets=None,
           uniforms=None, nthreads=0, want_stats=True, tuned=False):
    """Render frames [frame_begin, frame_begin+spp) -> (accum[h,w,4] float32, Stats).  tuned=True: the -O3 build (CPU baseline only)."""
    L = tuned_lib() if tuned else lib()
    rows = h - row0 if rows is None else rows
    u = uniforms if uniforms is not None else make_uniforms(w, h)
    off = offsets if offsets is not None else pixel_offsets(w, h)
    if accum is None:
        accum = np.zeros((h, w, 4), np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.shape == (h, w, 4)
    st = Stats()
    L.orc_render(scene.p, C.byref(u), off.ctypes.data, frame_begin, spp, bounces, row0, rows,
                 accum.ctypes.data, C.byref(st) if want_stats else None, nthreads)
    return accum, st


def raygen(w, h, frame_index=0, offsets=None, uniforms=None):
    L = lib()
    u = uniforms if uniforms is not None else make_uniforms(w, h, frame_index)
    u.frameIndex = frame_index
    off = offsets if offsets is not None else pixel_offsets(w, h)
    rays = np.zeros(w * h, RAY_DTYPE)
    dst = np.ones((h, w, 4), np.float32)
    for y in range(h):
        for x in range(w):
            L.orc_raygen(C.byref(u), off.ctypes.data, rays.ctypes.data, dst.ctypes.data, x, y)
    return rays


def intersect_nearest(scene, rays, brute=False):
    out = np.zeros(rays.shape[0], ISECT_DTYPE)
    if brute:
        lib().orc_intersect_nearest_brute(scene.p, rays.ctypes.data, rays.shape[0], out.ctypes.data)
    else:
        lib().orc_intersect_nearest(scene.p, rays.ctypes.data, rays.shape[0], out.ctypes.data, None)
    return out


def intersect_any(scene, rays, brute=False):
    out = np.zeros(rays.shape[0], np.float32)
    if brute:
        lib().orc_intersect_any_brute(scene.p, rays.ctypes.data, rays.shape[0], out.ctypes.data)
    else:
        lib().orc_intersect_any(scene.p, rays.ctypes.data, rays.shape[0], out.ctypes.data, None)
    return out


def nearest_f64(scene, rays):
    n = rays.shape[0]
    prim = np.empty(n, np.int32)
    t = np.empty(n, np.float64)
    margin = np.empty(n, np.float64)
    tt, mm = C.c_double(), C.c_double()
    L = lib()
    base = rays.ctypes.data
    for i in range(n):
        prim[i] = L.orc_nearest_f64(scene.p, base + i * 48, C.byref(tt), C.byref(mm))
        t[i], margin[i] = tt.value, mm.value
    return prim, t, margin


def postprocess(accum, flip_y=True):
    h, w, _ = accum.shape
    out = np.empty((h, w, 4), np.uint8)
    lib().orc_postprocess(np.ascontiguousarray(accum).ctypes.data, w, h, out.ctypes.data, 1 if flip_y else 0)
    return out


def set_trig_mode(mode):
    lib().orc_set_trig_mode(mode)


VAR_BOUNCE_SEES_LIGHT, VAR_LIGHT_NORMALISED = 1, 2


def set_variant(flags):
    """Negative controls of the radiometric screenshot pin (trg_oracle.h orc_set_variant); 0 = the reference's estimator."""
    lib().orc_set_variant(int(flags))


def num_threads():
    return int(lib().orc_num_threads())
