"""One smoke test per experiment that was moved out of the shipped library (round-4 verdict, item 7): the workgroup path pool and the
wavefront schedule live in experiments/ and are compiled into experiments/lib/libtoyraygun_hip_exp.so only (experiments/build.py).  The test
runs in a child process, because the ctypes binding of a process loads ONE library (TRG_HIP_SO selects the experimental one)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_SO = os.path.join(ROOT, "experiments", "lib", "libtoyraygun_hip_exp.so")
W8_SO = os.path.join(ROOT, "experiments", "lib", "libtoyraygun_hip_w8.so")

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import pyoracle as O
from toyraygun_amd import capi
assert capi.has_experiments(), capi.HIP_SO
kernel, w, h, spp, bnc = int(sys.argv[1]), 96, 64, 4, 3
scene = O.OracleScene.cornell_box()
b = scene.buffers()
off = O.pixel_offsets(w, h)
O.set_trig_mode(O.TRIG_PORTABLE)
ref, st = O.render(scene, w, h, spp, bnc, offsets=off)
c = capi.Context(w, h)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
c.set_pixel_offsets(off)
c.set_option(capi.OPT_STRICT, 1)
c.set_option(capi.OPT_KERNEL, kernel)
for force_global in (0, 1):          # the scene staged in LDS, and traversed from HBM
    c.set_option(capi.OPT_FORCE_GLOBAL, force_global)
    c.reset_stats()
    c.render(0, spp, bnc)
    img, gs = c.read_accum(), c.stats()
    assert gs.last_kernel == kernel, (gs.last_kernel, kernel)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "experimental schedule %%d differs from the oracle (force_global %%d)" %% (kernel, force_global)
    assert gs.rays == st.rays, (gs.rays, st.rays)
c.close()
print("ok", kernel)
"""


CHILD_W8 = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import pyoracle as O
from toyraygun_amd import capi
w, h, spp, bnc = 80, 48, 3, 3
for name, scene in (("cornell box kept in HBM", O.OracleScene.cornell_box()), ("lattice of 2,628 triangles", O.OracleScene.cornell_lattice(6))):
    b = scene.buffers()
    off = O.pixel_offsets(w, h)
    O.set_trig_mode(O.TRIG_PORTABLE)
    ref, st = O.render(scene, w, h, spp, bnc, offsets=off)
    c = capi.Context(w, h)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
    c.set_pixel_offsets(off)
    c.set_option(capi.OPT_STRICT, 1)
    c.set_option(capi.OPT_FORCE_GLOBAL, 1)
    for regen in (0, 1):             # lock step and path regeneration, both through trav_step_wide8
        c.set_option(capi.OPT_REGEN, regen)
        c.reset_stats()
        c.render(0, spp, bnc)
        img, gs = c.read_accum(), c.stats()
        assert gs.scene_in_lds == 0 and gs.last_regen == regen
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "8-wide traversal differs from the oracle: %%s, regen %%d" %% (name, regen)
        assert gs.rays == st.rays, (gs.rays, st.rays)
    c.close()
print("ok w8")
"""


@pytest.mark.gpu
def test_compressed_8wide_nodes_are_bit_exact():
    """GPU: the compressed 8-wide tree (experiments/trg_wide8.inc.h; profiles/r05/c4_wide8_experiment.md: built, correct, slower) -- the Cornell
    box kept in HBM and a 2,628-triangle lattice, strict build, lock step and path regeneration: the oracle's image and ray counts bit for bit."""
    assert os.path.exists(W8_SO), "experiments/lib/libtoyraygun_hip_w8.so is missing: python experiments/build.py"
    env = dict(os.environ, TRG_HIP_SO=W8_SO)
    p = subprocess.run([sys.executable, "-c", CHILD_W8 % {"root": ROOT}], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "ok w8" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])


CHILD_TAIL_FLAT = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import pyoracle as O
from toyraygun_amd import capi
from tests.util import image_metrics, TOL_RMSE, TOL_FRAC
what = sys.argv[1]
assert capi.has_experiments()
scene = O.OracleScene.cornell_box()
b = scene.buffers()
w, h, spp, bnc = 120, 80, 5, 6
off = O.pixel_offsets(w, h)
c = capi.Context(w, h)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
c.set_pixel_offsets(off)
if what == "rtail":      # the refilling tail kernel: strict build, bit for bit, ray counts included
    O.set_trig_mode(O.TRIG_PORTABLE)
    ref, st = O.render(scene, w, h, spp, bnc, offsets=off)
    c.set_option(capi.OPT_STRICT, 1)
    c.set_option(capi.OPT_FRAME_SPLIT, 1)     # (a small image would take frame lanes, which have no tail)
    c.set_option(capi.OPT_TAIL_BOUNCE, 2)
    c.set_option(capi.OPT_TAIL_REFILL, 1)
    c.reset_stats()
    c.render(0, spp, bnc)
    gs = c.stats()
    assert gs.last_tail_bounce == 2
    assert np.array_equal(c.read_accum().view(np.uint32), ref.view(np.uint32)) and gs.rays == st.rays
else:                    # the flat primitive list (TRG_FLAT_PRIMS=1 in the environment): shipped build, the stated tolerance
    ref, st = O.render(scene, w, h, spp, bnc, offsets=off)
    c.render(0, spp, bnc)
    rmse, frac_ok, _ = image_metrics(c.read_accum(), ref)
    assert rmse <= TOL_RMSE and frac_ok >= TOL_FRAC, (rmse, frac_ok)
    rng = np.random.default_rng(9)
    rays = np.zeros(20000, O.RAY_DTYPE)
    rays["origin"] = rng.uniform((-0.95, 0.05, -0.95), (0.95, 1.9, 3.0), (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3))
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["mask"] = rng.choice([1, 2, 3], 20000).astype(np.uint32)
    rays["maxDistance"] = np.where(rng.random(20000) < 0.25, rng.uniform(0.05, 3.0, 20000), np.inf).astype(np.float32)
    got, want = c.trace(rays), O.intersect_nearest(scene, rays)
    assert (got["primitiveIndex"] != want["primitiveIndex"]).mean() < 3e-3
    assert ((c.trace(rays, any_hit=True) >= 0) != (O.intersect_any(scene, rays) >= 0)).mean() < 2e-3
c.close()
print("ok", what)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["rtail", "flat"])
def test_round5_experiments(what):
    """GPU: the tail kernel with in-wave refill (bit-exact, slower: profiles/r05/c3_tail_refill_experiment.md) and the flat primitive list for tiny
    scenes (in tolerance; C2 slower, C3 faster: profiles/r05/c2_flat_list_experiment.md), both in the experiments library only."""
    env = dict(os.environ, TRG_HIP_SO=EXP_SO)
    if what == "flat":
        env["TRG_FLAT_PRIMS"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD_TAIL_FLAT % {"root": ROOT}, what], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and ("ok " + what) in p.stdout, (what, p.stdout[-1500:], p.stderr[-1500:])


def test_experimental_library_is_built_and_complete():
    """CPU: the experimental build exists next to the product library and exports every symbol of include/trg.h (no compute call)."""
    import ctypes
    from toyraygun_amd import capi
    assert os.path.exists(EXP_SO), "experiments/lib/libtoyraygun_hip_exp.so is missing: python experiments/build.py (or __graft_entry__.build())"
    lib = ctypes.CDLL(EXP_SO)
    for name in capi.SYMBOL_NAMES:
        assert hasattr(lib, name), name
    lib.trg_library_experiments.restype = ctypes.c_int
    assert lib.trg_library_experiments() == 1
    assert capi.load().trg_library_experiments() == (1 if os.environ.get("TRG_HIP_SO") == EXP_SO else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,name", [(1, "path pool (render_pool_kernel)"), (2, "wavefront schedule (wf_* kernels)")])
def test_experimental_schedule_is_bit_exact(kernel, name):
    """GPU: Cornell box 96 x 64, 4 spp, 3 bounces through the experimental schedule, strict build, LDS- and HBM-resident: the oracle's image and
    ray counts bit for bit."""
    env = dict(os.environ, TRG_HIP_SO=EXP_SO)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(kernel)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and ("ok %d" % kernel) in p.stdout, (name, p.stdout[-1500:], p.stderr[-1500:])
