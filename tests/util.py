"""Shared helpers for the test-suite (metrics, context construction)."""
import numpy as np


def image_metrics(img, ref):
    """Per-pixel L2 over RGB on the linear float4 buffer (SURVEY 8d 'Parity tolerance')."""
    d = img[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)
    l2 = np.sqrt((d ** 2).sum(-1))
    refn = np.sqrt((ref[..., :3].astype(np.float64) ** 2).sum(-1))
    rmse = float(np.sqrt((d ** 2).mean()))
    frac_ok = float((l2 <= 1e-4 * np.maximum(1.0, refn)).mean())
    return rmse, frac_ok, float(l2.max())


# Tolerance for the FAST (FMA-contracted) build against the oracle, from north_star / SURVEY 8d:
# RMSE <= 1e-3 and >= 99.9 % of pixels with L2 <= 1e-4 * max(1, |ref|).
TOL_RMSE = 1e-3
TOL_FRAC = 0.999
# C4 exception ("C4 tolerance"): the replicated-mesh scene has 85,184 small cubes, i.e. ~1 M silhouette
# and crease edges where the Cornell box has ~50.  A ray that grazes an edge may resolve to the other face under FMA
# contraction (an "edge flip": a whole different path, not a rounding difference), and the share of such pixels scales
# with the edge length in the picture: measured 0.15 % at 1 spp, against 0.002 % on the Cornell box.  The RMSE bar is
# unchanged; the pixel share for C4 is 99.7 %.
TOL_FRAC_C4 = 0.997


def make_ctx(O, scene, w, h, offsets=None, uniforms=None):
    from toyraygun_amd import capi
    c = capi.Context(w, h)
    b = scene.buffers()
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(O.uniforms_bytes(uniforms if uniforms is not None else O.make_uniforms(w, h)))
    if offsets is None:
        c.set_pixel_offsets_seed()
    else:
        c.set_pixel_offsets(offsets)
    return c
