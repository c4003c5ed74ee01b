"""C4 (1.02 M triangles, HBM-resident) steady-state ms per step vs launches in flight: python scripts/gpu_overlap_c4.py [depth]"""
import os, sys, time, subprocess; sys.path.insert(0, ".")
if len(sys.argv) == 1:
    for d in (1, 2, 4):
        subprocess.run([sys.executable, __file__, str(d)], env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
    sys.exit(0)
import numpy as np, torch
from toyraygun_amd import capi, host
from toyraygun_amd.dist import DistributedRenderer
depth = int(sys.argv[1])
W, H = 1920, 1080
b = host.Scene.cornell_lattice(44).buffers()
r = DistributedRenderer(W, H, 0, pipelined=depth > 1, depth=depth)
r.load_scene(b); r.ctx.set_uniforms(host.uniforms(W, H)[0]); r.ctx.set_pixel_offsets_seed()
r.ctx.render(0, 16, 3); ref = r.ctx.read_accum().copy()
r.ctx.set_option(capi.OPT_TIMING, 0)
for _ in range(4):
    r.render(0, 16, 3, gather=False)
r.synchronize()
K = 12
t0 = time.perf_counter()
for _ in range(K):
    f = r.render(0, 16, 3, gather=False)
r.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
same = all(np.array_equal(fr.cpu().numpy().view(np.uint32), ref.view(np.uint32)) for fr in r.frames)
print("%d in flight: %.2f ms per step, every frame buffer bit-identical to a launch alone: %s" % (depth, ms, same), flush=True)
