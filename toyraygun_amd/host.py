"""ctypes binding of libtoyraygun.so: the host-side C++ mirror of the reference's plugin surface
(Engine / Renderer / Scene / Shader / HipRenderer, include/engine/*.h) through its flat test shim
(toyraygun_amd/csrc/host/capi_host.cpp).  Scene building and camera math run on the CPU; anything that
renders needs the GPU library and a gfx950 device.
"""
import ctypes as C
import os

import numpy as np

from . import capi

HOST_SO = os.path.join(capi.LIB_DIR, "libtoyraygun.so")
_lib = None

SYMBOL_NAMES = ["trh_scene_new", "trh_scene_cornell", "trh_scene_lattice", "trh_scene_free", "trh_scene_add", "trh_scene_add_mesh", "trh_scene_add_mesh_colors", "trh_scene_add_obj", "trh_texture_load", "trh_texture_from_rgba",
                "trh_texture_info", "trh_texture_copy", "trh_texture_free", "trh_scene_add_textured_mesh", "trh_scene_add_obj_textured", "trh_scene_copy_textures", "trh_scene_texture_handles",
                "trh_scene_counts", "trh_scene_copy", "trh_mtx_srt", "trh_mtx_inverse", "trh_uniforms",
                "trh_random_texture", "trh_run_app", "trh_async_camera_move", "trh_async_camera_move_on", "trh_render_scene", "trh_render_scene_on"]


def load():
    global _lib
    if _lib is None:
        capi.load()  # libtoyraygun.so depends on libtoyraygun_hip.so
        if not os.path.exists(HOST_SO):
            raise ImportError("%s is missing: run __graft_entry__.build()" % HOST_SO)
        L = C.CDLL(HOST_SO)
        P, F = C.c_void_p, C.c_void_p
        L.trh_scene_new.restype = P
        L.trh_scene_cornell.restype = P
        L.trh_scene_lattice.restype = P
        L.trh_scene_lattice.argtypes = [C.c_int, C.c_uint32]
        L.trh_scene_free.argtypes = [P]
        L.trh_scene_add.argtypes = [P, C.c_int, F, F]
        L.trh_scene_add_mesh.argtypes = [P, F, F, F, C.c_int, F, F, C.c_uint]
        L.trh_scene_add_mesh_colors.argtypes = [P, F, F, F, F, C.c_int, F, C.c_uint]
        L.trh_texture_load.argtypes = [C.c_char_p]
        L.trh_texture_load.restype = P
        L.trh_texture_from_rgba.argtypes = [F, C.c_int, C.c_int]
        L.trh_texture_from_rgba.restype = P
        L.trh_texture_info.argtypes = [P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.trh_texture_copy.argtypes = [P, F]
        L.trh_texture_free.argtypes = [P]
        L.trh_scene_add_textured_mesh.argtypes = [P, F, F, F, F, C.c_int, F, F, C.c_uint, P]
        L.trh_scene_add_obj_textured.argtypes = [P, C.c_char_p, F, F, C.c_uint, P]
        L.trh_scene_add_obj_textured.restype = C.c_int
        L.trh_scene_copy_textures.argtypes = [P, F, F]
        L.trh_scene_texture_handles.restype = C.c_uint
        L.trh_scene_texture_handles.argtypes = [P, C.POINTER(C.c_void_p), C.c_uint]
        L.trh_scene_add_obj.argtypes = [P, C.c_char_p, F, F, C.c_uint]
        L.trh_scene_add_obj.restype = C.c_int
        L.trh_scene_counts.argtypes = [P, C.POINTER(C.c_uint)]
        L.trh_scene_counts.restype = C.c_uint
        L.trh_scene_copy.argtypes = [P, F, F, F, F, F]
        L.trh_mtx_srt.argtypes = [F, F, F, F]
        L.trh_mtx_inverse.argtypes = [F, F]
        L.trh_uniforms.argtypes = [C.c_int, C.c_int, C.c_int, F, F, F, F]
        L.trh_random_texture.argtypes = [C.c_int, C.c_int, C.c_uint32, F]
        L.trh_run_app.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, F, C.c_char_p,
                                  C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)]
        L.trh_run_app.restype = C.c_int
        L.trh_async_camera_move.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, F, C.c_int, C.c_int, F, C.POINTER(C.c_uint)]
        L.trh_async_camera_move.restype = C.c_int
        L.trh_render_scene.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, F, C.POINTER(C.c_double)]
        L.trh_render_scene.restype = C.c_int
        L.trh_async_camera_move_on.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, F, C.c_int, C.c_int, C.c_void_p, C.c_int, F, C.POINTER(C.c_uint)]
        L.trh_async_camera_move_on.restype = C.c_int
        L.trh_render_scene_on.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, F, C.POINTER(C.c_double)]
        L.trh_render_scene_on.restype = C.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class Scene:
    """toyraygun::Scene (include/engine/Scene.h)."""
    KINDS = {"cube": 0, "plane": 1, "light": 2}

    def __init__(self, handle=None):
        self.L = load()
        self.h = handle if handle is not None else self.L.trh_scene_new()
        self._textures = []   # Texture objects the native Scene borrows

    @classmethod
    def cornell_box(cls):
        return cls(load().trh_scene_cornell())

    @classmethod
    def cornell_lattice(cls, n, seed=0x5EED0002):
        """BASELINE config C4: Cornell box + n^3 replicated cubes (n = 44 -> 1,022,244 triangles)."""
        return cls(load().trh_scene_lattice(int(n), seed))

    def __del__(self):
        try:
            self.L.trh_scene_free(self.h)
        except Exception:
            pass

    def add(self, kind, color, mtx):
        c, m = _f32(color), _f32(mtx).reshape(16)
        self.L.trh_scene_add(self.h, self.KINDS[kind], c.ctypes.data, m.ctypes.data)

    def add_mesh(self, verts, normals, tri_idx, mtx, color, material_id):
        """Scene::addMesh.  color: one RGB for the whole mesh, or an [n_verts, 3] array of per-vertex colours."""
        v, n, m, c = _f32(verts), _f32(normals), _f32(mtx).reshape(16), _f32(color)
        t = np.ascontiguousarray(tri_idx, np.uint32)
        if c.ndim == 2:
            assert c.shape == v.reshape(-1, 3).shape
            self.L.trh_scene_add_mesh_colors(self.h, v.ctypes.data, n.ctypes.data, c.ctypes.data, t.ctypes.data, t.size // 3,
                                             m.ctypes.data, material_id)
            return
        self.L.trh_scene_add_mesh(self.h, v.ctypes.data, n.ctypes.data, t.ctypes.data, t.size // 3, m.ctypes.data,
                                  c.ctypes.data, material_id)

    def add_obj(self, path, mtx, color, material_id=1, texture=None):
        """Scene::addObj: returns the number of triangles added (-1: unreadable file)."""
        m, c = _f32(mtx).reshape(16), _f32(color)
        if texture is not None:
            self._textures.append(texture)
            return int(self.L.trh_scene_add_obj_textured(self.h, str(path).encode(), m.ctypes.data, c.ctypes.data, material_id, texture.h))
        return int(self.L.trh_scene_add_obj(self.h, str(path).encode(), m.ctypes.data, c.ctypes.data, material_id))

    def add_textured_mesh(self, verts, normals, uvs, tri_idx, mtx, color, material_id, texture):
        """Scene::addMesh with texture coordinates [n_verts, 2] and an albedo Texture."""
        v, n, uv, m, c = _f32(verts), _f32(normals), _f32(uvs), _f32(mtx).reshape(16), _f32(color)
        t = np.ascontiguousarray(tri_idx, np.uint32)
        self._textures.append(texture)   # the Scene borrows it
        self.L.trh_scene_add_textured_mesh(self.h, v.ctypes.data, n.ctypes.data, uv.ctypes.data, t.ctypes.data, t.size // 3, m.ctypes.data,
                                           c.ctypes.data, material_id, texture.h)

    def texture_buffers(self):
        """(uvs [n_verts, 2], texture_ids [n_tris], images list of [h, w, 4] uint8 in texture-id order) for trg_load_textures."""
        nt = C.c_uint()
        nv = self.L.trh_scene_counts(self.h, C.byref(nt))
        uv, ids = np.zeros((nv, 2), np.float32), np.zeros(nt.value, np.uint32)
        self.L.trh_scene_copy_textures(self.h, uv.ctypes.data, ids.ctypes.data)
        # the image list follows the NATIVE registration order (Scene::textureID registers a texture when a face with texture
        # coordinates first uses it -- an OBJ without `vt`, or an unreadable file, registers nothing), not the order Python saw them in
        n = self.L.trh_scene_texture_handles(self.h, None, 0)
        handles = (C.c_void_p * max(n, 1))()
        self.L.trh_scene_texture_handles(self.h, handles, n)
        by_handle = {int(t.h): t for t in self._textures}
        imgs = [by_handle[int(handles[k])].rgba() for k in range(n)]
        return uv, ids, imgs

    def buffers(self):
        nt = C.c_uint()
        nv = self.L.trh_scene_counts(self.h, C.byref(nt))
        nt = nt.value
        out = dict(positions=np.zeros((nv, 3), np.float32), normals=np.zeros((nv, 3), np.float32),
                   colors=np.zeros((nv, 3), np.float32), indices=np.zeros(nv, np.uint32),
                   material_ids=np.zeros(nt, np.uint32))
        self.L.trh_scene_copy(self.h, out["positions"].ctypes.data, out["normals"].ctypes.data, out["colors"].ctypes.data,
                              out["indices"].ctypes.data, out["material_ids"].ctypes.data)
        return out


class Texture:
    """toyraygun::Texture: loadFile (8-bit PNG / binary PPM) or raw RGBA8 pixels."""

    def __init__(self, path=None, rgba=None):
        self.L = load()
        if path is not None:
            self.h = self.L.trh_texture_load(str(path).encode())
            if not self.h:
                raise IOError("Texture::loadFile failed for %s" % path)
        else:
            a = np.ascontiguousarray(rgba, np.uint8)
            assert a.ndim == 3 and a.shape[2] == 4
            self.h = self.L.trh_texture_from_rgba(a.ctypes.data, a.shape[1], a.shape[0])

    def info(self):
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        self.L.trh_texture_info(self.h, C.byref(w), C.byref(h), C.byref(c))
        return w.value, h.value, c.value

    def pixels(self):
        w, h, c = self.info()
        out = np.zeros((h, w, c), np.uint8)
        self.L.trh_texture_copy(self.h, out.ctypes.data)
        return out

    def rgba(self):
        p = self.pixels()
        h, w, c = p.shape
        out = np.full((h, w, 4), 255, np.uint8)
        if c >= 3:
            out[..., :3] = p[..., :3]
            if c == 4:
                out[..., 3] = p[..., 3]
        else:
            out[..., :3] = p[..., :1]
            if c == 2:
                out[..., 3] = p[..., 1]
        return out

    def __del__(self):
        try:
            if self.h:
                self.L.trh_texture_free(self.h)
        except Exception:
            pass


def mtx_srt(scale, rot, pos):
    m = np.zeros(16, np.float32)
    s, r, t = _f32(scale), _f32(rot), _f32(pos)  # keep the temporaries alive across the call
    load().trh_mtx_srt(m.ctypes.data, s.ctypes.data, r.ctypes.data, t.ctypes.data)
    return m


def mtx_inverse(m):
    out = np.zeros(16, np.float32)
    src = _f32(m).reshape(16)
    load().trh_mtx_inverse(out.ctypes.data, src.ctypes.data)
    return out


def uniforms(w, h, frame_index=0, eye=(0.0, 1.0, 3.38), at=(0.0, 1.0, -1.0)):
    """(176-byte Uniforms block, viewProj[16]) as HipRenderer::fillUniforms builds them."""
    raw = np.zeros(176, np.uint8)
    vp = np.zeros(16, np.float32)
    e, a = _f32(eye), _f32(at)
    load().trh_uniforms(w, h, frame_index, e.ctypes.data, a.ctypes.data, raw.ctypes.data, vp.ctypes.data)
    return capi.Uniforms.from_buffer_copy(raw.tobytes()), vp


def random_texture(w, h, seed=capi.SEED_OFFSETS):
    out = np.zeros(w * h, np.uint32)
    load().trh_random_texture(w, h, seed, out.ctypes.data)
    return out


def _device_list(devices):
    d = np.ascontiguousarray(devices if devices is not None else [], np.int32)
    return d, (d.ctypes.data if d.size else None), int(d.size)


def async_camera_move(w, h, frames_a, frames_b, eye_b, bounces=3, device=0, devices=None):
    """HipRenderer: frames_a x renderFrame(), setCameraPosition(eye_b), frames_b x renderFrame(); returns (accum, launches).
    devices: a list of HIP devices -> HipRenderer::setDevices (a device group, also for one device)."""
    acc = np.zeros((h, w, 4), np.float32)
    e = _f32(eye_b)
    n = C.c_uint()
    d, dp, dn = _device_list(devices)
    rc = load().trh_async_camera_move_on(w, h, frames_a, frames_b, e.ctypes.data, bounces, device, dp, dn, acc.ctypes.data, C.byref(n))
    if rc != 0:
        raise RuntimeError("trh_async_camera_move failed at step %d" % rc)
    return acc, n.value


def render_scene(scene, w, h, frames, bounces=3, device_build=0, device=0, devices=None):
    """HipRenderer on a host.Scene: setDeviceBuild(device_build), loadScene, `frames` samples; returns (accum[h,w,4], load ms).
    devices: as in async_camera_move."""
    acc = np.zeros((h, w, 4), np.float32)
    ms = C.c_double()
    d, dp, dn = _device_list(devices)
    rc = load().trh_render_scene_on(scene.h, w, h, frames, bounces, device_build, device, dp, dn, acc.ctypes.data, C.byref(ms))
    if rc != 0:
        raise RuntimeError("trh_render_scene failed at step %d" % rc)
    return acc, ms.value


def run_app(w, h, frames, bounces=3, batch=False, device=0, png_path=None, want_launches=False):
    """Drive the reference app's call sequence headlessly; returns (accum[h,w,4], ms, rays[, launches]): ms = wall time of
    the frame loop from the first renderFrame() to the completion of the last frame on the device."""
    acc = np.zeros((h, w, 4), np.float32)
    ms, rays, launches = C.c_double(), C.c_ulonglong(), C.c_uint()
    rc = load().trh_run_app(w, h, frames, bounces, 1 if batch else 0, device, acc.ctypes.data,
                            png_path.encode() if png_path else None, C.byref(ms), C.byref(rays), C.byref(launches))
    if rc != 0:
        raise RuntimeError("trh_run_app failed at step %d" % rc)
    return (acc, ms.value, rays.value, launches.value) if want_launches else (acc, ms.value, rays.value)
