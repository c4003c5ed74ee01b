"""Where should the LDS-resident path hand over to the HBM path?  Cornell box + n random triangles, LDS vs forced HBM."""
import sys; sys.path.insert(0, ".")
import numpy as np
from toyraygun_amd import capi, host
W, H = 1920, 1080
rng = np.random.default_rng(5)
for n in (0, 40, 80, 120, 160, 200, 240):
    s = host.Scene.cornell_box()
    if n:
        ctr = rng.uniform([-0.8, 0.2, -0.8], [0.8, 1.8, 0.8], (n, 3)).astype(np.float32)
        tri = ctr[:, None, :] + rng.normal(0, 0.08, (n, 3, 3)).astype(np.float32)
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-9)
        s.add_mesh(tri.reshape(-1, 3), np.repeat(nrm, 3, axis=0).astype(np.float32), np.arange(3 * n, dtype=np.uint32), np.eye(4, dtype=np.float32), (0.6, 0.6, 0.6), 1)
    b = s.buffers()
    c = capi.Context(W, H)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    out = []
    for fg in (0, 1):
        c.set_option(capi.OPT_FORCE_GLOBAL, fg)
        ts = []
        for _ in range(4):
            c.render(0, 16, 3); ts.append(c.stats().last_render_ms)
        st = c.stats()
        out.append("%s %.2f ms (lds %d B)" % ("HBM" if fg else "LDS" if st.scene_in_lds else "hbm", min(ts), st.lds_bytes))
    print("tris %d scene %d B | %s" % (b["material_ids"].shape[0], st.scene_bytes, " | ".join(out)), flush=True)
    c.close()
