"""C2 kernel time vs BVH builder knobs (env TRG_BVH_MAXLEAF / TRG_BVH_TRAVCOST), one process per setting."""
import os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    from toyraygun_amd import capi, host
    W, H = 1920, 1080
    b = host.Scene.cornell_box().buffers()
    c = capi.Context(W, H)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    out = []
    for bn in (3, 8):
        ts = []
        for _ in range(6):
            c.render(0, 16, bn); ts.append(c.stats().last_render_ms)
        out.append("b%d %.3f" % (bn, min(ts)))
    st = c.stats()
    print("maxleaf=%s travcost=%s nodes=%d depth=%d : %s" % (os.environ.get("TRG_BVH_MAXLEAF"), os.environ.get("TRG_BVH_TRAVCOST"), st.bvh_nodes, st.bvh_depth, " ".join(out)), flush=True)
    sys.exit(0)
for ml, tc in (("2", "1.2"), ("1", "1.2"), ("1", "0.5"), ("2", "0.5"), ("2", "0.8"), ("3", "1.2"), ("4", "1.2"), ("2", "2.0"), ("3", "0.8"), ("2", "1.2")):
    env = dict(os.environ, TRG_BVH_MAXLEAF=ml, TRG_BVH_TRAVCOST=tc)
    subprocess.run([sys.executable, __file__, "--one"], env=env)
