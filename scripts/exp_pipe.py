"""Steady-state ms per step (4 frames in flight) of C2 and 1/8-band with experimental libraries, one process each:
python scripts/exp_pipe.py <exp name> ..."""
import os, sys, time, subprocess
sys.path.insert(0, ".")
if sys.argv[1].startswith("--one="):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    from toyraygun_amd import capi
    name = sys.argv[1][6:]
    if name != "shipped":
        capi.HIP_SO = os.path.join("exp_build", name, "libtoyraygun_hip.so")
    from toyraygun_amd import host
    from toyraygun_amd.dist import DistributedRenderer, band_rows
    W, H = 1920, 1080
    b = host.Scene.cornell_box().buffers()
    r = DistributedRenderer(W, H, 0, pipelined=True)
    r.load_scene(b); r.ctx.set_uniforms(host.uniforms(W, H)[0]); r.ctx.set_pixel_offsets_seed()
    r.ctx.set_option(capi.OPT_TIMING, 0)
    out = []
    for n, bn in ((1, 3), (8, 3), (1, 8)):
        r.row0, r.rows = band_rows(H, n, n // 2)
        for _ in range(8):
            r.render(0, 16, bn, gather=False)
        r.synchronize()
        K = 40 if bn == 3 else 15
        t0 = time.perf_counter()
        for _ in range(K):
            r.render(0, 16, bn, gather=False)
        r.synchronize()
        out.append("1/%d b%d %.3f" % (n, bn, (time.perf_counter() - t0) / K * 1e3))
    print("%-10s %s" % (name, " | ".join(out)), flush=True)
    sys.exit(0)
for rep in range(2):
    for n in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one=" + n])
