"""Quick on-GPU sanity run (developer script): HIP path vs the CPU oracle on C1, timing on C2."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")  # run from the repo root: python tests/dev/<script>.py
from oracle import pyoracle as O
from toyraygun_amd import capi

def ctx_for(scene, w, h, offsets=None):
    c = capi.Context(w, h)
    b = scene.buffers()
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
    if offsets is None: c.set_pixel_offsets_seed()
    else: c.set_pixel_offsets(offsets)
    return c

s = O.OracleScene.cornell_box()
w = h = 256
off = O.pixel_offsets(w, h)
c = ctx_for(s, w, h, off)
print("stats", {k: v for k, v in c.stats().as_dict().items() if k.startswith(("bvh", "scene", "lds"))})
# halton bit-exactness
rng = np.random.default_rng(1)
i = rng.integers(0, 2**32, 200000, dtype=np.uint64).astype(np.uint32); d = rng.integers(0, 64, 200000).astype(np.uint32)
g = c.halton(i, d); ref = np.array([O.halton(a, b) for a, b in zip(i[:20000], d[:20000])], np.float32)
print("halton mismatches", int((g[:20000].view(np.uint32) != ref.view(np.uint32)).sum()))
for trig, strict in ((O.TRIG_PORTABLE, 1), (O.TRIG_LIBM, 0)):
    O.set_trig_mode(trig)
    for (spp, bnc) in ((1, 1), (4, 3)):
        ref_img, st = O.render(s, w, h, spp, bnc, offsets=off)
        c.set_option(capi.OPT_STRICT, strict); c.reset_stats()
        c.render(0, spp, bnc)
        img = c.read_accum(); gs = c.stats()
        diff = np.abs(img[..., :3] - ref_img[..., :3])
        nbad = int((diff.max(-1) > 1e-5).sum())
        print(f"strict={strict} spp={spp} b={bnc}: max|d|={diff.max():.3e} rmse={np.sqrt((diff**2).mean()):.3e} px>1e-5={nbad} "
              f"bitexact={bool((img.view(np.uint32)==ref_img.view(np.uint32)).all())} rays gpu={gs.rays} cpu={st.rays} ms={gs.last_render_ms:.3f}")
c.close()
# C2 timing
w, h = 1920, 1080
c = ctx_for(s, w, h)
for it in range(3):
    c.reset_stats(); c.render(0, 16, 3); st = c.stats()
    print(f"C2 1920x1080 16spp 3b: {st.last_render_ms:.3f} ms rays={st.rays} -> {st.rays/st.last_render_ms/1e3:.1f} Mrays/s")
c.set_option(capi.OPT_COUNTERS, 1); c.reset_stats(); c.render(0, 16, 3); st = c.stats()
print("counters:", st.as_dict())
from PIL import Image
Image.fromarray(c.postprocess()).save("gpurun_out/c2.png")
