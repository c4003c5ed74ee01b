import sys, os, ctypes as C, numpy as np, subprocess
sys.path.insert(0, ".")
from toyraygun_amd import capi
W, H = 1920, 1080
if sys.argv[1] == "--dump":
    from toyraygun_amd import host
    for n, tag in ((0, "c2"), (44, "c4")):
        b = host.Scene.cornell_lattice(n).buffers() if n else host.Scene.cornell_box().buffers()
        np.savez("/tmp/exp_%s.npz" % tag, u=np.frombuffer(bytes(host.uniforms(W, H)[0]), np.uint8), **b)
    sys.exit(0)
if sys.argv[1].startswith("--one="):
    name = sys.argv[1][6:]
    capi.HIP_SO = os.path.join("exp_build", name, "libtoyraygun_hip.so")
    out = []
    for tag in ("c2", "c4"):
        z = np.load("/tmp/exp_%s.npz" % tag)
        c = capi.Context(W, H)
        c.load_scene(z["positions"], z["normals"], z["colors"], z["indices"], z["material_ids"])
        c.set_uniforms(z["u"].tobytes()); c.set_pixel_offsets_seed()
        for bn in (3, 8) if tag == "c2" else (3,):
            ts = []
            for i in range(4):
                c.reset_stats(); c.render(0, 16, bn); st = c.stats(); ts.append(st.last_render_ms)
            out.append("%s b%d %.3f ms (%.2f Grays/s)" % (tag, bn, min(ts), st.rays / min(ts) / 1e6))
        c.close()
    print("%-8s %s" % (name, " | ".join(out)))
    sys.exit(0)
subprocess.run([sys.executable, __file__, "--dump"], check=True)
for rep in range(2):
    for n in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one=" + n])
