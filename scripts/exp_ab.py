"""A/B timing of experimental builds of libtoyraygun_hip.so (scripts/exp_build.sh <name> -D...), one process per variant:
python scripts/exp_ab.py <config: c2|c3|c4|c2b8> <variant|shipped> ...  -> launch-alone ms (min of 5) and steady-state ms per
step with four frames in flight, rays from the in-kernel counters."""
import os, sys, time, subprocess
sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
CFG = {"c4xl": (1920, 1080, 16, 3, "lattice96"), "c2": (1920, 1080, 16, 3, "box"), "c3": (1920, 1080, 64, 8, "box"), "c4": (1920, 1080, 16, 3, "lattice"), "c2b8": (1920, 1080, 16, 8, "box"),
       "c2b4": (1920, 1080, 16, 4, "box"), "c2b5": (1920, 1080, 16, 5, "box"), "c2b6": (1920, 1080, 16, 6, "box")}
CFG.update({"mesh%d" % k: (1920, 1080, 16, 3, "sphere%d" % k) for k in range(2, 9)})   # Cornell box + a smooth-shaded icosphere of 20 * 4^k triangles (no quads to pair)


from scenes import make_scene   # noqa: E402  (scripts/scenes.py)
if sys.argv[1].startswith("--one="):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    from toyraygun_amd import capi
    cfg, name = sys.argv[1][6:].split(":")
    if name != "shipped":
        capi.HIP_SO = os.path.join("exp_build", name, "libtoyraygun_hip.so")
    from toyraygun_amd import host
    from toyraygun_amd.dist import DistributedRenderer
    W, H, spp, bn, sc = CFG[cfg]
    b = make_scene(host, sc).buffers()
    r = DistributedRenderer(W, H, 0, pipelined=True)
    for k, v in [kv.split("=") for kv in os.environ.get("TRG_EXP_OPTS", "").split(",") if kv]:
        r.ctx.set_option(int(k), int(v))      # (before the scene too: option 6 picks the builder)
    r.load_scene(b); r.ctx.set_uniforms(host.uniforms(W, H)[0]); r.ctx.set_pixel_offsets_seed()
    for k, v in [kv.split("=") for kv in os.environ.get("TRG_EXP_OPTS", "").split(",") if kv]:
        r.ctx.set_option(int(k), int(v))
    r.ctx.set_option(capi.OPT_TIMING, 1)
    r.ctx.reset_stats(); r.ctx.render(0, spp, bn); rays = r.ctx.stats().rays
    alone = []
    for _ in range(5):
        r.ctx.render(0, spp, bn); alone.append(r.ctx.stats().last_render_ms)
    r.ctx.set_option(capi.OPT_TIMING, 0)
    for _ in range(6):
        r.render(0, spp, bn, gather=False)
    r.synchronize()
    K = 30 if min(alone) < 5 else 10
    t0 = time.perf_counter()
    for _ in range(K):
        r.render(0, spp, bn, gather=False)
    r.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    print("%-4s %-14s alone %.3f ms (%.2f Grays/s) | pipelined %.3f ms/step (%.2f Grays/s)" % (cfg, name, min(alone), rays / min(alone) / 1e6, ms, rays / ms / 1e6), flush=True)
    sys.exit(0)
cfg = sys.argv[1]
for rep in range(2):
    for n in sys.argv[2:]:
        subprocess.run([sys.executable, __file__, "--one=%s:%s" % (cfg, n)])
