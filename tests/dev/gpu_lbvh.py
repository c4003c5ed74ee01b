"""GPU LBVH build (TRG_OPT_GPU_BUILD) vs host SAH build: bit-exact images (strict), build time, traversal speed."""
import sys, time; sys.path.insert(0, ".")  # run from the repo root: python tests/dev/<script>.py
import numpy as np
from oracle import pyoracle as O
from toyraygun_amd import capi, host
for n in (0, 6, 44):
    osc = O.OracleScene.cornell_lattice(n) if n else O.OracleScene.cornell_box()
    b = osc.buffers()
    w = h = 128
    off = O.pixel_offsets(w, h)
    O.set_trig_mode(O.TRIG_PORTABLE); ref, rst = O.render(osc, w, h, 2, 3, offsets=off); O.set_trig_mode(O.TRIG_LIBM)
    for gpu in (0, 1):
        c = capi.Context(w, h)
        c.set_option(capi.OPT_GPU_BUILD, gpu)
        t0 = time.time(); c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"]); wall = time.time() - t0
        c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h))); c.set_pixel_offsets(off)
        c.set_option(capi.OPT_STRICT, 1)
        res = []
        for k in (0, 1):
            c.set_option(capi.OPT_KERNEL, k); c.reset_stats(); c.render(0, 2, 3)
            res.append(bool(np.array_equal(c.read_accum().view(np.uint32), ref.view(np.uint32))) and c.stats().rays == rst.rays)
        st = c.stats()
        print(f"n={n} tris={b['material_ids'].shape[0]} gpu_build={gpu}: build {st.last_build_ms:.2f} ms (load wall {wall*1e3:.0f} ms) nodes4={st.bvh_nodes4} depth4={st.bvh_depth4} in_lds={st.scene_in_lds} bitexact={res}")
        c.close()
# traversal speed on C4 with each tree
W, H = 1920, 1080
b = host.Scene.cornell_lattice(44).buffers()
for gpu in (0, 1):
    c = capi.Context(W, H); c.set_option(capi.OPT_GPU_BUILD, gpu)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    ts = []
    for i in range(3):
        c.reset_stats(); c.render(0, 16, 3); st = c.stats(); ts.append(st.last_render_ms)
    print(f"C4 gpu_build={gpu}: build {st.last_build_ms:.2f} ms, render {min(ts):.2f} ms, {st.rays/min(ts)/1e6:.2f} Grays/s, depth4 {st.bvh_depth4}, lds {st.lds_bytes}")
    c.close()
