"""Times C2 with experimental library variants: python scripts/exp_time.py name1 name2 ... (A/B interleaved)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, ".")
from toyraygun_amd import capi
names = sys.argv[1:]
W, H, SPP, B = 1920, 1080, 16, 3
if names and names[0] == "--dump":
    from toyraygun_amd import host
    b = host.Scene.cornell_box().buffers()
    np.savez("/tmp/exp_scene.npz", u=np.frombuffer(bytes(host.uniforms(W, H)[0]), np.uint8), **b)
    sys.exit(0)
res = {}
def run(name, rounds=5):
    # each variant in a fresh process-wide library handle
    capi._lib = None
    capi.HIP_SO = os.path.join("exp_build", name, "libtoyraygun_hip.so") if name != "shipped" else os.path.join(capi.LIB_DIR, "libtoyraygun_hip.so")
    L = C.CDLL(capi.HIP_SO)
    for n, r, a in capi._SYMBOLS:
        f = getattr(L, n); f.restype = r; f.argtypes = a
    capi._lib = L
    z = np.load("/tmp/exp_scene.npz"); buf = z; u = z["u"].tobytes()
    c = capi.Context(W, H)
    c.load_scene(buf["positions"], buf["normals"], buf["colors"], buf["indices"], buf["material_ids"])
    c.set_uniforms(u); c.set_pixel_offsets_seed()
    out = []
    for kern in (0, 1):
        c.set_option(capi.OPT_KERNEL, kern)
        for bn in (1, 3, 8):
            ts = []
            for i in range(rounds + 1):
                c.reset_stats(); c.render(0, SPP, bn); st = c.stats()
                if i >= 1: ts.append(st.last_render_ms)
            out.append("k%d b%d %.3f" % (kern, bn, min(ts)))
    print("%-10s lds=%d  %s" % (name, st.lds_bytes, "  ".join(out)))
    c.close()
    return [1.0], st.rays
import subprocess
if len(names) == 1 and names[0].startswith("--one="):
    n = names[0][6:]; ts, rays = run(n)
    pass
else:
    subprocess.run([sys.executable, __file__, "--dump"], check=True)
    for rep in range(1):
        for n in names:
            subprocess.run([sys.executable, __file__, "--one=" + n])
