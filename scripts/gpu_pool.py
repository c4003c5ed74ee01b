import sys
sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
b = host.Scene.cornell_box().buffers()
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
for bn in (3, 8):
    for k in (0, 1):
        c.set_option(capi.OPT_KERNEL, k)
        ts = []
        for _ in range(5):
            c.render(0, 16, bn); ts.append(c.stats().last_render_ms)
        print("bounces %d kernel %s: %.3f ms" % (bn, "pool" if k else "direct", min(ts)))
