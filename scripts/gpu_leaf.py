"""Kernel time vs the host BVH builder's knobs (env TRG_BVH_MAXLEAF / TRG_BVH_TRAVCOST), one process per setting:
python scripts/gpu_leaf.py        C2 (Cornell box in LDS), 3 and 8 bounces
python scripts/gpu_leaf.py c4     the 1,022,244-triangle lattice (regeneration kernel): time, boxes and triangle tests per ray"""
import os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 1 and sys.argv[1] == "--one4":
    from toyraygun_amd import capi, host
    W, H = 1920, 1080
    b = host.Scene.cornell_lattice(44).buffers()
    c = capi.Context(W, H)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    c.set_option(capi.OPT_COUNTERS, 1); c.reset_stats(); c.render(0, 16, 3); st = c.stats()
    c.set_option(capi.OPT_COUNTERS, 0)
    ts = []
    for _ in range(4):
        c.render(0, 16, 3); ts.append(c.stats().last_render_ms)
    print("maxleaf=%s travcost=%s wide nodes=%d depth4=%d : %.2f ms, %.1f boxes/ray, %.2f triangle tests/ray" % (os.environ.get("TRG_BVH_MAXLEAF"), os.environ.get("TRG_BVH_TRAVCOST"),
          st.bvh_nodes4, st.bvh_depth4, min(ts), 2.0 * st.node_fetches / st.rays, st.tri_tests / st.rays), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "c4":
    for ml, tc in (("2", "1.2"), ("1", "1.2"), ("3", "1.2"), ("4", "1.2"), ("2", "0.6"), ("2", "2.0"), ("4", "2.0"), ("4", "0.6")):
        subprocess.run([sys.executable, __file__, "--one4"], env=dict(os.environ, TRG_BVH_MAXLEAF=ml, TRG_BVH_TRAVCOST=tc))
    sys.exit(0)
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
for ml, tc in (("2", "1.2"), ("1", "1.2"), ("1", "0.5"), ("2", "0.5"), ("2", "0.8"), ("3", "1.2"), ("4", "1.2"), ("2", "2.0"), ("3", "0.8"), ("4", "3.0"), ("6", "3.0"), ("8", "3.0"), ("6", "8.0"), ("8", "8.0"), ("2", "1.2")):
    env = dict(os.environ, TRG_BVH_MAXLEAF=ml, TRG_BVH_TRAVCOST=tc)
    subprocess.run([sys.executable, __file__, "--one"], env=env)
