"""C4 workload for profiling: python scripts/c4_run.py [launches]   (TRG_EXP_OPTS="option=value,..." as scripts/exp_ab.py; the options are set
before the scene is loaded, so 6=1 selects the device SAH builder; TRG_RUN_SCENE=sphere5 / lattice96 / box: another scene of scripts/scenes.py)"""
import os, sys; sys.path.insert(0, "."); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from toyraygun_amd import capi, host
from scenes import make_scene
W, H = 1920, 1080
b = make_scene(host, os.environ.get("TRG_RUN_SCENE", "lattice44")).buffers()
c = capi.Context(W, H)
for k, v in [kv.split("=") for kv in os.environ.get("TRG_EXP_OPTS", "").split(",") if kv]:
    c.set_option(int(k), int(v))
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    c.reset_stats(); c.render(0, 16, 3); st = c.stats()
    print("C4 %.2f ms %d rays %.2f Grays/s" % (st.last_render_ms, st.rays, st.rays / st.last_render_ms / 1e6))
