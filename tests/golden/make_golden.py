"""Generates tests/golden/golden_v1.npz from the CPU oracle (oracle/trg_oracle.c).

The reference has no tests, golden vectors or runnable CPU path, so these vectors pin
the ORACLE (and, on the GPU box, the HIP path) against regressions; they are not reference outputs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import pyoracle as O  # noqa: E402

DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")


def halton_inputs():
    rng = np.random.default_rng(0x5EED)
    special = list(range(0, 64)) + [2 ** 24 - 1, 2 ** 24, 2 ** 24 + 1, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 2, 2 ** 32 - 1]
    rnd = rng.integers(0, 2 ** 32, 57, dtype=np.uint64).tolist()
    return np.array(special + rnd, np.uint32)


def sampling_inputs():
    g = np.linspace(0.0, 0.999, 7, dtype=np.float32)
    r = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    rng = np.random.default_rng(7)
    n = rng.normal(size=(r.shape[0], 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)
    p = rng.uniform(-1, 1, size=(r.shape[0], 3)).astype(np.float32) + np.array([0, 1, 0], np.float32)
    return r, n.astype(np.float32), p.astype(np.float32)


def sampling_outputs(r, n, p, u):
    import ctypes as C
    L = O.lib()
    fp = C.POINTER(C.c_float)
    hemi = np.zeros((r.shape[0], 3), np.float32)
    aligned = np.zeros((r.shape[0], 3), np.float32)
    ldir = np.zeros((r.shape[0], 3), np.float32)
    lcol = np.zeros((r.shape[0], 3), np.float32)
    ldist = np.zeros(r.shape[0], np.float32)
    for k in range(r.shape[0]):
        rr, nn, pp = r[k].copy(), n[k].copy(), p[k].copy()
        L.orc_sample_cosine_hemisphere(rr.ctypes.data_as(fp), hemi[k].ctypes.data_as(fp))
        h = hemi[k].copy()
        L.orc_align_hemisphere(h.ctypes.data_as(fp), nn.ctypes.data_as(fp), aligned[k].ctypes.data_as(fp))
        d = C.c_float()
        L.orc_sample_area_light(C.byref(u), rr.ctypes.data_as(fp), pp.ctypes.data_as(fp), nn.ctypes.data_as(fp),
                                ldir[k].ctypes.data_as(fp), lcol[k].ctypes.data_as(fp), C.byref(d))
        ldist[k] = d.value
    return hemi, aligned, ldir, lcol, ldist


def main():
    out = {}
    s = O.OracleScene.cornell_box()
    for k, v in s.buffers().items():
        out["scene_" + k] = v
    hi = halton_inputs()
    out["halton_i"] = hi
    out["halton"] = np.array([[O.halton(i, d) for d in range(64)] for i in hi], np.float32)
    u256 = O.make_uniforms(256, 256)
    out["uniforms_256"] = np.frombuffer(O.uniforms_bytes(u256), np.uint8)
    out["uniforms_1920x1080"] = np.frombuffer(O.uniforms_bytes(O.make_uniforms(1920, 1080)), np.uint8)
    out["offsets_16x16"] = O.pixel_offsets(16, 16)
    r, n, p = sampling_inputs()
    out["sample_r"], out["sample_n"], out["sample_p"] = r, n, p
    for mode, tag in ((O.TRIG_LIBM, "libm"), (O.TRIG_PORTABLE, "portable")):
        O.set_trig_mode(mode)
        hemi, aligned, ldir, lcol, ldist = sampling_outputs(r, n, p, u256)
        out["sample_hemi_" + tag], out["sample_aligned_" + tag] = hemi, aligned
        if mode == O.TRIG_LIBM:
            out["sample_ldir"], out["sample_lcol"], out["sample_ldist"] = ldir, lcol, ldist
        img, st = O.render(s, 64, 64, 4, 3)
        out["frame_64_4spp_3b_" + tag] = img
        out["rays_64_4spp_3b_" + tag] = np.array([st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits], np.uint64)
        img, st = O.render(s, 48, 32, 2, 8)
        out["frame_48x32_2spp_8b_" + tag] = img
    O.set_trig_mode(O.TRIG_LIBM)
    for f in (0, 5):
        out["raygen_16x16_f%d" % f] = O.raygen(16, 16, f).view(np.uint8)
    img, st = O.render(s, 256, 256, 1, 1)  # C1 (independent of the trig mode: the bounce ray is never traced)
    out["frame_c1_256_1spp_1b"] = img
    out["rays_c1"] = np.array([st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits], np.uint64)
    np.savez_compressed(DST, **out)
    print("wrote", DST, os.path.getsize(DST), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
