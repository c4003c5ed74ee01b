"""WHOLE-frame parity of the full-size configurations against the oracle -- what scripts/gpu_fullframe.py prints, as a test (VERDICT r03,
item 2): every pixel of C2, C3 (2.97 G rays), C4 (1,022,244 triangles) and C5 (3840x2160), the strict build bit for bit including the
three ray counts, the shipped build within the stated tolerance.  About seven minutes of oracle time on the GPU box's host threads for all four:
C4 runs in every `-m gpu` suite (round 5), the others are opt-in:  python -m pytest tests -m "gpu and slow"   (the plain suite compares
sampled rows of C3 / C5 and the whole frame of C2)."""
import numpy as np
import pytest

from tests.util import TOL_FRAC, TOL_FRAC_C4, TOL_RMSE, image_metrics

pytestmark = [pytest.mark.gpu]

CONFIGS = {"c2": (1920, 1080, 16, 3, "box"), "c3": (1920, 1080, 256, 8, "box"), "c4": (1920, 1080, 16, 3, "lattice"), "c5": (3840, 2160, 64, 3, "box")}


# C4 -- the configuration with the loosened pixel-share bar (tests/util.py TOL_FRAC_C4) -- is compared as a WHOLE frame in every `-m gpu` run
# (round-4 verdict, item 6c: about 90 s of oracle time on the box's 16 host threads); C2 (also covered whole by test_full_size_c2_properties),
# C3 and C5 stay opt-in: `-m "gpu and slow"`.
@pytest.mark.parametrize("name", [pytest.param("c2", marks=pytest.mark.slow), pytest.param("c3", marks=pytest.mark.slow), "c4", pytest.param("c5", marks=pytest.mark.slow)])
def test_whole_frame_parity(built, O, name):
    from toyraygun_amd import capi
    w, h, spp, bnc, kind = CONFIGS[name]
    scene = O.OracleScene.cornell_lattice(44) if kind == "lattice" else O.OracleScene.cornell_box()
    b = scene.buffers()
    off = O.pixel_offsets(w, h)
    c = capi.Context(w, h)
    try:
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
        c.set_pixel_offsets(off)
        O.set_trig_mode(O.TRIG_PORTABLE)
        try:
            ref, rst = O.render(scene, w, h, spp, bnc, offsets=off)
        finally:
            O.set_trig_mode(O.TRIG_LIBM)
        c.set_option(capi.OPT_STRICT, 1)
        c.reset_stats()
        c.render(0, spp, bnc)
        img, st = c.read_accum(), c.stats()
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "%s strict: %d pixels differ" % (name, int((img.view(np.uint32) != ref.view(np.uint32)).any(-1).sum()))
        assert (st.primary_rays, st.bounce_rays, st.shadow_rays) == (rst.primary_rays, rst.bounce_rays, rst.shadow_rays)
        ref, _ = O.render(scene, w, h, spp, bnc, offsets=off)
        c.set_option(capi.OPT_STRICT, 0)
        c.render(0, spp, bnc)
        rmse, frac_ok, worst = image_metrics(c.read_accum(), ref)
        assert rmse <= TOL_RMSE and frac_ok >= (TOL_FRAC_C4 if name == "c4" else TOL_FRAC), (name, rmse, frac_ok, worst)
    finally:
        c.close()
