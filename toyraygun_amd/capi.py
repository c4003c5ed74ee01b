"""ctypes binding of the C ABI in include/trg.h (libtoyraygun_hip.so).

This is the only way Python reaches the renderer: there is no Python or CPU fallback.  Loading fails
loudly (ImportError/OSError) when the HIP library has not been built, and trg_create fails loudly
when no gfx950 device is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HIP_SO = os.environ.get("TRG_HIP_SO") or os.path.join(LIB_DIR, "libtoyraygun_hip.so")   # TRG_HIP_SO: an experimental build (scripts/exp_build.sh)

OK = 0
ERR_INVALID, ERR_NOMEM, ERR_DEVICE, ERR_NODEV, ERR_RANGE = -22, -12, -5, -19, -34
OPT_STRICT, OPT_COUNTERS, OPT_FORCE_GLOBAL, OPT_TIMING, OPT_KERNEL, OPT_GPU_BUILD, OPT_FRAME_SPLIT, OPT_LAUNCHES_IN_FLIGHT = 1, 2, 3, 4, 5, 6, 7, 8
OPT_TAIL_BOUNCE, OPT_TAIL_LEVELS, OPT_REGEN, OPT_TILE_ORDER, OPT_STACK_LDS_LEVELS, OPT_TAIL_SORT, OPT_TAIL_REFILL = 9, 10, 11, 12, 13, 14, 15
KERNEL_DIRECT, KERNEL_POOL, KERNEL_WAVEFRONT, KERNEL_AUTO = 0, 1, 2, -1
MATERIAL_DEFAULT, MATERIAL_EMISSIVE = 1, 2
MAX_BOUNCES = 15
SEED_OFFSETS = 0x5EED0001


class Uniforms(C.Structure):
    """trg_uniforms: the 176-byte shader Uniforms block (runtime/shaders/common.h:14-34)."""
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("frameIndex", C.c_uint32), ("_pad", C.c_uint32),
        ("cam_pos", C.c_float * 4),
        ("inv_view_proj", C.c_float * 16),
        ("light_pos", C.c_float * 4), ("light_forward", C.c_float * 4), ("light_right", C.c_float * 4),
        ("light_up", C.c_float * 4), ("light_color", C.c_float * 4),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("shaded_hits", C.c_uint64),
        ("node_fetches", C.c_uint64), ("tri_tests", C.c_uint64), ("wave_node_iters", C.c_uint64), ("wave_tri_iters", C.c_uint64),
        ("last_render_ms", C.c_double), ("total_render_ms", C.c_double),
        ("renders", C.c_uint32), ("bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("bvh_leaves", C.c_uint32),
        ("scene_in_lds", C.c_uint32), ("lds_bytes", C.c_uint32),
        ("scene_bytes", C.c_uint64),
        ("last_build_ms", C.c_double),
        ("gpu_built", C.c_uint32), ("bvh_nodes4", C.c_uint32), ("bvh_depth4", C.c_uint32), ("last_frame_split", C.c_uint32),
        ("last_tail_bounce", C.c_uint32), ("last_kernel", C.c_uint32), ("last_regen", C.c_uint32), ("last_tile_order", C.c_uint32),
        ("bvh_quads", C.c_uint32), ("bvh_boxes", C.c_uint32),
    ]

    @property
    def rays(self):
        return self.primary_rays + self.bounce_rays + self.shadow_rays

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("mask", "<u4"), ("direction", "<f4", 3),
                      ("maxDistance", "<f4"), ("color", "<f4", 4)])
ISECT_DTYPE = np.dtype([("distance", "<f4"), ("primitiveIndex", "<i4"), ("coordinates", "<f4", 2)])
assert C.sizeof(Uniforms) == 176 and RAY_DTYPE.itemsize == 48 and ISECT_DTYPE.itemsize == 16

# every symbol include/trg.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SYMBOLS = [
    ("trg_create", C.c_int, [C.POINTER(_P), C.c_int, C.c_uint32, C.c_uint32]),
    ("trg_destroy", None, [_P]),
    ("trg_last_error", C.c_char_p, [_P]),
    ("trg_library_experiments", C.c_int, []),
    ("trg_load_scene", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32]),
    ("trg_load_textures", C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(_P), _P, _P, C.c_uint32]),
    ("trg_set_uniforms", C.c_int, [_P, C.POINTER(Uniforms)]),
    ("trg_set_pixel_offsets", C.c_int, [_P, _P]),
    ("trg_set_pixel_offsets_seed", C.c_int, [_P, C.c_uint32]),
    ("trg_render", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("trg_read_accum", C.c_int, [_P, _P]),
    ("trg_get_stats", C.c_int, [_P, C.POINTER(Stats)]),
    ("trg_reset_stats", C.c_int, [_P]),
    ("trg_set_option", C.c_int, [_P, C.c_int, C.c_int64]),
    ("trg_bind_accum", C.c_int, [_P, _P]),
    ("trg_accum_device_ptr", C.c_int, [_P, C.POINTER(_P)]),
    ("trg_set_stream", C.c_int, [_P, _P]),
    ("trg_sync", C.c_int, [_P]),
    ("trg_stream_idle", C.c_int, [_P]),
    ("trg_fence_record", C.c_int, [_P, C.c_int]),
    ("trg_fence_wait", C.c_int, [_P, C.c_int]),
    ("trg_band_rows", None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("trg_group_create", C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.c_int, C.c_uint32, C.c_uint32]),
    ("trg_group_destroy", None, [_P]),
    ("trg_group_last_error", C.c_char_p, [_P]),
    ("trg_group_size", C.c_int, [_P]),
    ("trg_group_exchange", C.c_int, [_P]),
    ("trg_group_ctx", _P, [_P, C.c_int]),
    ("trg_group_load_scene", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32]),
    ("trg_group_load_textures", C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(_P), _P, _P, C.c_uint32]),
    ("trg_group_set_uniforms", C.c_int, [_P, C.POINTER(Uniforms)]),
    ("trg_group_set_pixel_offsets_seed", C.c_int, [_P, C.c_uint32]),
    ("trg_group_set_option", C.c_int, [_P, C.c_int, C.c_int64]),
    ("trg_group_render", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    ("trg_group_sync", C.c_int, [_P]),
    ("trg_group_fence_record", C.c_int, [_P, C.c_int]),
    ("trg_group_fence_wait", C.c_int, [_P, C.c_int]),
    ("trg_group_read_accum", C.c_int, [_P, C.c_int, _P]),
    ("trg_group_get_stats", C.c_int, [_P, C.POINTER(Stats)]),
    ("trg_group_set_pipeline", C.c_int, [_P, C.c_int]),
    ("trg_group_pipeline_depth", C.c_int, [_P]),
    ("trg_group_time_launches", C.c_int, [_P, C.c_int]),
    ("trg_group_launch_ms", C.c_int, [_P, C.c_int, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("trg_group_exchange_note", C.c_char_p, [_P]),
    ("trg_group_rccl_info", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    ("trg_debug_plane_records", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, C.c_uint32, C.POINTER(C.c_uint32), _P]),
    ("trg_group_set_bands", C.c_int, [_P, C.c_int]),
    ("trg_group_bands", C.c_int, [_P]),
    ("trg_group_postprocess", C.c_int, [_P, C.c_int, _P, C.c_int]),
    ("trg_microband_rows", None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("trg_render_bands", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("trg_unpack_bands", C.c_int, [_P, _P, _P, C.c_uint32]),
    ("trg_trace", C.c_int, [_P, _P, C.c_size_t, C.c_int, _P]),
    ("trg_halton", C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    ("trg_halton_table", C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    ("trg_raygen", C.c_int, [_P, C.c_uint32, _P]),
    ("trg_sample", C.c_int, [_P, _P, _P, _P, C.c_size_t, _P]),
    ("trg_postprocess", C.c_int, [_P, _P, C.c_int]),
    ("trg_debug_build_bvh", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("trg_debug_build_bvh4", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("trg_debug_scene_layout", C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), _P]),
    ("trg_debug_xcc_ids", C.c_int, [_P, C.c_uint32, _P]),
    ("trg_debug_leaf_records", C.c_int, [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("trg_debug_tile_of_slot", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("trg_debug_build_bvh4q", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("trg_debug_boxes", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
]
SYMBOL_NAMES = [s[0] for s in _SYMBOLS]

_lib = None


def load():
    """dlopen libtoyraygun_hip.so and bind every symbol of include/trg.h.  No fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_SO):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  toyraygun_amd has no CPU fallback." % HIP_SO)
        L = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        for name, res, args in _SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def has_experiments():
    """True when the loaded library is experiments/lib/libtoyraygun_hip_exp.so (TRG_HIP_SO): it has the path-pool and wavefront schedules."""
    return bool(load().trg_library_experiments())


class TrgError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "trg error %d: %s" % (code, msg))
        self.code = code


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Context:
    """One rendering context on one GPU (trg_ctx)."""

    def __init__(self, width, height, device=0):
        self.L = load()
        self.w, self.h = int(width), int(height)
        self.device = device
        h = _P()
        rc = self.L.trg_create(C.byref(h), device, self.w, self.h)
        if rc != OK:
            raise TrgError(rc, (self.L.trg_last_error(None) or b"").decode())
        self.h_ctx = h

    def close(self):
        if getattr(self, "h_ctx", None):
            self.L.trg_destroy(self.h_ctx)
            self.h_ctx = None

    __del__ = close

    def _chk(self, rc):
        if rc != OK:
            raise TrgError(rc, (self.L.trg_last_error(self.h_ctx) or b"").decode())

    # ---- scene / uniforms / offsets ----
    def load_scene(self, positions, normals, colors, indices, material_ids):
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        col = np.ascontiguousarray(colors, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        mat = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
        nt = mat.shape[0]
        if idx.shape[0] != 3 * nt or nrm.shape[0] != 3 * nt or col.shape[0] != 3 * nt:
            raise ValueError("expected 3*n_tris indices/normals/colors for %d triangles" % nt)
        self._chk(self.L.trg_load_scene(self.h_ctx, _ptr(pos), _ptr(nrm), _ptr(col), _ptr(idx), _ptr(mat), pos.shape[0], nt))

    def load_textures(self, uvs, texture_ids, images):
        """trg_load_textures: uvs [3*n_tris, 2], texture_ids [n_tris] (0 = none, k = images[k-1]), images = list of [h, w, 4] uint8."""
        uv = np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(texture_ids, np.uint32).reshape(-1)
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        for im in imgs:
            if im.ndim != 3 or im.shape[2] != 4:
                raise ValueError("textures must be [h, w, 4] uint8")
        ptrs = (_P * max(len(imgs), 1))(*[im.ctypes.data for im in imgs])
        ws = np.array([im.shape[1] for im in imgs], np.uint32)
        hs = np.array([im.shape[0] for im in imgs], np.uint32)
        self._chk(self.L.trg_load_textures(self.h_ctx, _ptr(uv), _ptr(ids), ids.shape[0], ptrs, _ptr(ws), _ptr(hs), len(imgs)))

    def set_uniforms(self, u):
        if not isinstance(u, Uniforms):
            raw = bytes(u) if not isinstance(u, (bytes, bytearray)) else u
            if len(raw) != 176:
                raise ValueError("uniforms must be 176 bytes")
            u = Uniforms.from_buffer_copy(raw)
        self._chk(self.L.trg_set_uniforms(self.h_ctx, C.byref(u)))

    def set_pixel_offsets(self, offsets):
        off = np.ascontiguousarray(offsets, np.uint32).reshape(-1)
        if off.shape[0] != self.w * self.h:
            raise ValueError("offsets must have width*height entries")
        self._chk(self.L.trg_set_pixel_offsets(self.h_ctx, _ptr(off)))

    def set_pixel_offsets_seed(self, seed=SEED_OFFSETS):
        self._chk(self.L.trg_set_pixel_offsets_seed(self.h_ctx, seed))

    def set_option(self, opt, value):
        self._chk(self.L.trg_set_option(self.h_ctx, opt, int(value)))

    # ---- the hot path ----
    def render(self, frame_begin, spp, bounces, row0=0, rows=None):
        rows = self.h - row0 if rows is None else rows
        self._chk(self.L.trg_render(self.h_ctx, frame_begin, spp, bounces, row0, rows))

    def render_bands(self, frame_begin, spp, bounces, n_ranks, rank, store_row0):
        """trg_render_bands: the 8-row micro-bands rank, rank + n_ranks, ... of the image, stored compactly from row store_row0 of the
        bound accumulation buffer (microband_rows gives the rows and the ranks' common stride)."""
        self._chk(self.L.trg_render_bands(self.h_ctx, frame_begin, spp, bounces, n_ranks, rank, store_row0))

    def unpack_bands(self, compact_ptr, image_ptr, n_ranks):
        """trg_unpack_bands: compact frame of n_ranks interleaved bands (device pointer) -> image rows (device pointer), on the context's stream."""
        self._chk(self.L.trg_unpack_bands(self.h_ctx, C.c_void_p(int(compact_ptr)), C.c_void_p(int(image_ptr)), n_ranks))

    def read_accum(self):
        out = np.empty((self.h, self.w, 4), np.float32)
        self._chk(self.L.trg_read_accum(self.h_ctx, _ptr(out)))
        return out

    def stats(self):
        st = Stats()
        self._chk(self.L.trg_get_stats(self.h_ctx, C.byref(st)))
        return st

    def reset_stats(self):
        self._chk(self.L.trg_reset_stats(self.h_ctx))

    def sync(self):
        self._chk(self.L.trg_sync(self.h_ctx))

    def stream_idle(self):
        rc = self.L.trg_stream_idle(self.h_ctx)
        if rc < 0:
            self._chk(rc)
        return bool(rc)

    def fence_record(self, slot):
        self._chk(self.L.trg_fence_record(self.h_ctx, slot))

    def fence_wait(self, slot):
        self._chk(self.L.trg_fence_wait(self.h_ctx, slot))

    def bind_accum(self, device_ptr):
        self._chk(self.L.trg_bind_accum(self.h_ctx, device_ptr))

    def accum_device_ptr(self):
        p = _P()
        self._chk(self.L.trg_accum_device_ptr(self.h_ctx, C.byref(p)))
        return p.value

    def set_stream(self, hip_stream):
        self._chk(self.L.trg_set_stream(self.h_ctx, hip_stream))

    def xcc_ids(self, n_blocks):
        """trg_debug_xcc_ids: the XCD (HW_REG_XCC_ID) every workgroup of a plain launch of n_blocks workgroups ran on."""
        out = np.zeros(int(n_blocks), np.uint32)
        self._chk(self.L.trg_debug_xcc_ids(self.h_ctx, int(n_blocks), _ptr(out)))
        return out

    # ---- stage-level entry points ----
    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        n = rays.shape[0]
        out = np.zeros(n, np.float32 if any_hit else ISECT_DTYPE)
        self._chk(self.L.trg_trace(self.h_ctx, _ptr(rays), n, 1 if any_hit else 0, _ptr(out)))
        return out

    def halton(self, i, d):
        i = np.ascontiguousarray(i, np.uint32).reshape(-1)
        d = np.ascontiguousarray(d, np.uint32).reshape(-1)
        out = np.zeros(i.shape[0], np.float32)
        self._chk(self.L.trg_halton(self.h_ctx, _ptr(i), _ptr(d), i.shape[0], _ptr(out)))
        return out

    def halton_table(self, i, d):
        """a5 through the shipped megakernel's own path (digit-group tables in LDS for dimensions 1..5)."""
        i = np.ascontiguousarray(i, np.uint32).reshape(-1)
        d = np.ascontiguousarray(d, np.uint32).reshape(-1)
        out = np.zeros(i.shape[0], np.float32)
        self._chk(self.L.trg_halton_table(self.h_ctx, _ptr(i), _ptr(d), i.shape[0], _ptr(out)))
        return out

    def raygen(self, frame_index):
        out = np.zeros(self.w * self.h, RAY_DTYPE)
        self._chk(self.L.trg_raygen(self.h_ctx, frame_index, _ptr(out)))
        return out

    def sample(self, p, n, r):
        p = np.ascontiguousarray(p, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(n, np.float32).reshape(-1, 3)
        r = np.ascontiguousarray(r, np.float32).reshape(-1, 4)
        out = np.zeros((p.shape[0], 12), np.float32)
        self._chk(self.L.trg_sample(self.h_ctx, _ptr(p), _ptr(n), _ptr(r), p.shape[0], _ptr(out)))
        return out

    def postprocess(self, flip_y=True):
        out = np.empty((self.h, self.w, 4), np.uint8)
        self._chk(self.L.trg_postprocess(self.h_ctx, _ptr(out), 1 if flip_y else 0))
        return out


GATHER_NONE, GATHER_ALL, GATHER_ROOT = 0, 1, 2
BANDS_CONTIGUOUS, BANDS_INTERLEAVED = 0, 1
EXCHANGE_NONE, EXCHANGE_RCCL, EXCHANGE_COPY = 0, 1, 2


def band_rows(height, n, rank):
    """trg_band_rows: rows [row0, row0 + rows) of device `rank` of `n` (bands of ceil(height / n) rows; host-only)."""
    a, b = C.c_uint32(), C.c_uint32()
    load().trg_band_rows(height, n, rank, C.byref(a), C.byref(b))
    return a.value, b.value


def microband_rows(height, n, rank):
    """trg_microband_rows: (rows of rank's interleaved band, the ranks' common stride in rows) -- 8-row micro-bands rank, rank + n, ..."""
    a, b = C.c_uint32(), C.c_uint32()
    load().trg_microband_rows(height, n, rank, C.byref(a), C.byref(b))
    return a.value, b.value


class Group:
    """trg_group: one context per device of one node, row-band sharding, RCCL exchange (include/trg.h)."""

    def __init__(self, devices, width, height):
        self.L = load()
        self.w, self.h = int(width), int(height)
        devs = (C.c_int * len(devices))(*devices)
        g = _P()
        rc = self.L.trg_group_create(C.byref(g), devs, len(devices), self.w, self.h)
        if rc != OK:
            raise TrgError(rc, (self.L.trg_group_last_error(None) or b"").decode())
        self.g = g
        self.n = len(devices)

    @property
    def exchange(self):
        """How the bands travel: EXCHANGE_NONE (one device), EXCHANGE_RCCL or EXCHANGE_COPY (TRG_GROUP_EXCHANGE=copy / no librccl.so)."""
        return self.L.trg_group_exchange(self.g)

    def close(self):
        if getattr(self, "g", None):
            self.L.trg_group_destroy(self.g)
            self.g = None

    __del__ = close

    def _chk(self, rc):
        if rc != OK:
            raise TrgError(rc, (self.L.trg_group_last_error(self.g) or b"").decode())

    def load_scene(self, positions, normals, colors, indices, material_ids):
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        col = np.ascontiguousarray(colors, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        mat = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
        self._chk(self.L.trg_group_load_scene(self.g, _ptr(pos), _ptr(nrm), _ptr(col), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0]))

    def load_textures(self, uvs, texture_ids, images):
        uv = np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(texture_ids, np.uint32).reshape(-1)
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        ptrs = (_P * max(len(imgs), 1))(*[im.ctypes.data for im in imgs])
        ws = np.array([im.shape[1] for im in imgs], np.uint32)
        hs = np.array([im.shape[0] for im in imgs], np.uint32)
        self._chk(self.L.trg_group_load_textures(self.g, _ptr(uv), _ptr(ids), ids.shape[0], ptrs, _ptr(ws), _ptr(hs), len(imgs)))

    def fence_record(self, slot):
        self._chk(self.L.trg_group_fence_record(self.g, slot))

    def fence_wait(self, slot):
        self._chk(self.L.trg_group_fence_wait(self.g, slot))

    def set_uniforms(self, u):
        if not isinstance(u, Uniforms):
            u = Uniforms.from_buffer_copy(bytes(u))
        self._chk(self.L.trg_group_set_uniforms(self.g, C.byref(u)))

    def set_pixel_offsets_seed(self, seed=SEED_OFFSETS):
        self._chk(self.L.trg_group_set_pixel_offsets_seed(self.g, seed))

    def set_option(self, opt, value):
        self._chk(self.L.trg_group_set_option(self.g, opt, int(value)))

    def render(self, frame_begin, spp, bounces, gather=GATHER_ALL, root=0):
        self._chk(self.L.trg_group_render(self.g, frame_begin, spp, bounces, gather, root))

    def sync(self):
        self._chk(self.L.trg_group_sync(self.g))

    def set_pipeline(self, depth):
        """`depth` independent frames in flight per device (trg_group_set_pipeline): depth frame buffers and render streams + one
        exchange stream per device; 1 = the progressive one-buffer mode."""
        self._chk(self.L.trg_group_set_pipeline(self.g, int(depth)))

    @property
    def pipeline_depth(self):
        return self.L.trg_group_pipeline_depth(self.g)

    @property
    def exchange_note(self):
        """Why the exchange is not the one asked for ("" when it is): the RCCL -> peer-copy fallback says so here."""
        return (self.L.trg_group_exchange_note(self.g) or b"").decode()

    def rccl_info(self):
        """trg_group_rccl_info: {"version": ncclGetVersion, "ranks": [ncclCommCount of every device's communicator]} -- what RCCL itself saw --,
        or None for a group without communicators (one device, the copy exchange)."""
        if self.exchange != EXCHANGE_RCCL:
            return None
        v = C.c_int(0)
        ranks = (C.c_int * self.n)()
        self._chk(self.L.trg_group_rccl_info(self.g, C.byref(v), ranks, self.n))
        return {"version": int(v.value), "ranks": [int(x) for x in ranks]}

    def set_bands(self, mode):
        """BANDS_CONTIGUOUS (rows [g*B, (g+1)*B) per device) or BANDS_INTERLEAVED (8-row micro-bands dealt round robin, trg_render_bands)."""
        self._chk(self.L.trg_group_set_bands(self.g, int(mode)))

    @property
    def bands(self):
        return self.L.trg_group_bands(self.g)

    def postprocess(self, rank=0, flip_y=True):
        out = np.empty((self.h, self.w, 4), np.uint8)
        self._chk(self.L.trg_group_postprocess(self.g, rank, _ptr(out), 1 if flip_y else 0))
        return out

    def time_launches(self, on=True):
        self._chk(self.L.trg_group_time_launches(self.g, 1 if on else 0))

    def launch_ms(self, rank, cap=4096):
        """Durations (ms) of rank's launches since the last read, from HIP events on the streams they ran on (waits for the rank)."""
        buf = (C.c_double * cap)()
        n = C.c_uint32(0)
        self._chk(self.L.trg_group_launch_ms(self.g, rank, buf, cap, C.byref(n)))
        return [buf[k] for k in range(min(n.value, cap))]

    def read_accum(self, rank=0):
        out = np.empty((self.h, self.w, 4), np.float32)
        self._chk(self.L.trg_group_read_accum(self.g, rank, _ptr(out)))
        return out

    def stats(self):
        st = Stats()
        self._chk(self.L.trg_group_get_stats(self.g, C.byref(st)))
        return st

    def reset_stats(self):
        for r in range(self.n):
            if self.L.trg_reset_stats(self.L.trg_group_ctx(self.g, r)) != OK:
                raise TrgError(ERR_INVALID, "trg_reset_stats failed for rank %d" % r)

    def rank_stats(self, rank):
        """trg_get_stats of one device's context (trg_group_ctx)."""
        st = Stats()
        ctx = self.L.trg_group_ctx(self.g, rank)
        if not ctx or self.L.trg_get_stats(ctx, C.byref(st)) != OK:
            raise TrgError(ERR_INVALID, "trg_group_ctx / trg_get_stats failed for rank %d" % rank)
        return st

    def rank_rays(self):
        return [int(self.rank_stats(r).rays) for r in range(self.n)]


def debug_plane_records(positions, indices, material_ids):
    """Host-only (no GPU): the plane form of the triangle records -- (planes[n,12] float32, meta[n] uint16, center[3] float32) in the record
    order of debug_build_bvh; the planes are relative to `center`."""
    L = load()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    mat = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
    n = C.c_uint32()
    ctr = np.zeros(3, np.float32)
    rc = L.trg_debug_plane_records(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], None, None, 0, C.byref(n), None)
    if rc != OK:
        raise TrgError(rc, "trg_debug_plane_records")
    planes, meta = np.zeros((n.value, 12), np.float32), np.zeros(n.value, np.uint16)
    rc = L.trg_debug_plane_records(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], _ptr(planes), _ptr(meta), n.value, C.byref(n), _ptr(ctr))
    if rc != OK:
        raise TrgError(rc, "trg_debug_plane_records")
    return planes, meta, ctr


def debug_build_bvh(positions, indices, material_ids):
    """Host-only BVH build (no GPU): returns (nodes[n,16] float32, tris[m,12] float32, depth)."""
    L = load()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    mat = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
    nn, nr, dp = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = L.trg_debug_build_bvh(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], None, 0, None, 0,
                               C.byref(nn), C.byref(nr), C.byref(dp))
    if rc != OK:
        raise TrgError(rc, "trg_debug_build_bvh")
    nodes = np.zeros((nn.value, 16), np.float32)
    tris = np.zeros((nr.value, 12), np.float32)
    rc = L.trg_debug_build_bvh(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], _ptr(nodes), nn.value,
                               _ptr(tris), nr.value, C.byref(nn), C.byref(nr), C.byref(dp))
    if rc != OK:
        raise TrgError(rc, "trg_debug_build_bvh")
    return nodes, tris, dp.value


def debug_boxes(positions, indices, material_ids):
    """Host-only: the box leaves of the scene, [n, 20] float32 (include/trg.h trg_debug_boxes)."""
    L = load()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    mat = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
    n = C.c_uint32()
    rc = L.trg_debug_boxes(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], None, 0, C.byref(n))
    if rc != OK:
        raise TrgError(rc, "trg_debug_boxes")
    out = np.zeros((n.value, 20), np.float32)
    if n.value:
        rc = L.trg_debug_boxes(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], _ptr(out), n.value, C.byref(n))
        if rc != OK:
            raise TrgError(rc, "trg_debug_boxes")
    return out


def debug_build_bvh4(positions, indices, material_ids):
    """Host-only 4-wide BVH (no GPU): returns (nodes4[n,32] float32, depth)."""
    L = load()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    mat = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
    nn, dp = C.c_uint32(), C.c_uint32()
    rc = L.trg_debug_build_bvh4(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], None, 0, C.byref(nn), C.byref(dp))
    if rc != OK:
        raise TrgError(rc, "trg_debug_build_bvh4")
    nodes = np.zeros((nn.value, 32), np.float32)
    rc = L.trg_debug_build_bvh4(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], _ptr(nodes), nn.value, C.byref(nn), C.byref(dp))
    if rc != OK:
        raise TrgError(rc, "trg_debug_build_bvh4")
    return nodes, dp.value


def debug_build_bvh4q(positions, indices, material_ids):
    """Host-only quantised 4-wide BVH (no GPU): nodes4q[n,16] uint32, node i = node i of debug_build_bvh4."""
    L = load()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    mat = np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
    nn = C.c_uint32()
    rc = L.trg_debug_build_bvh4q(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], None, 0, C.byref(nn))
    if rc != OK:
        raise TrgError(rc, "trg_debug_build_bvh4q")
    nodes = np.zeros((nn.value, 16), np.uint32)
    rc = L.trg_debug_build_bvh4q(_ptr(pos), _ptr(idx), _ptr(mat), pos.shape[0], mat.shape[0], _ptr(nodes), nn.value, C.byref(nn))
    if rc != OK:
        raise TrgError(rc, "trg_debug_build_bvh4q")
    return nodes
