"""Per-band kernel times with an experimental library: python scripts/gpu_bands2.py <exp name>"""
import sys, os, subprocess, numpy as np
sys.path.insert(0, ".")
W, H = 1920, 1080
if sys.argv[1] == "--dump":
    from toyraygun_amd import host
    b = host.Scene.cornell_box().buffers()
    np.savez("/tmp/exp_c2.npz", u=np.frombuffer(bytes(host.uniforms(W, H)[0]), np.uint8), **b); sys.exit(0)
if sys.argv[1].startswith("--one="):
    from toyraygun_amd import capi
    from toyraygun_amd.dist import band_rows
    name = sys.argv[1][6:]
    capi.HIP_SO = os.path.join("exp_build", name, "libtoyraygun_hip.so")
    z = np.load("/tmp/exp_c2.npz")
    c = capi.Context(W, H)
    c.load_scene(z["positions"], z["normals"], z["colors"], z["indices"], z["material_ids"])
    c.set_uniforms(z["u"].tobytes()); c.set_pixel_offsets_seed()
    out = []
    for n in (1, 2, 4, 8):
        ts = []
        for r in range(n):
            row0, rows = band_rows(H, n, r)
            best = 1e9
            for _ in range(4):
                c.render(0, 16, 3, row0, rows); best = min(best, c.stats().last_render_ms)
            ts.append(best)
        out.append("N=%d max %.3f" % (n, max(ts)))
    print("%-8s %s" % (name, " | ".join(out))); sys.exit(0)
subprocess.run([sys.executable, __file__, "--dump"], check=True)
for n in sys.argv[1:]:
    subprocess.run([sys.executable, __file__, "--one=" + n])
