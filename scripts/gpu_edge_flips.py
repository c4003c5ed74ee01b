"""Edge flips of the SHIPPED build as a function of edge density (round-4 verdict, item 6a).  A ray that passes within a few 1e-7 of a triangle edge may
resolve to the other face under FMA contraction / the plane-form test: a different path, not a rounding difference -- an "outlier" pixel of
tests/util.py's metric.  How many there are depends on how much edge a picture holds.  This script measures it:

  python scripts/gpu_edge_flips.py [seed]  ->  one line per case: triangles, image size, spp, bounces, lambda = projected triangle-edge length per pixel
                                                  (pixels of edge per pixel of image, every edge of every triangle, clipped to the image, occlusion ignored),
                                                  outlier pixels, outlier share, share / lambda;  then the fitted bound tests/util.py states.

Cases: the Cornell box + n random triangles (n = 0 ... 10,000; three size classes, as scripts/gpu_fuzz.py draws them) at 32^2 ... 256^2, 1 - 32 spp, 1 - 6
bounces, shipped build against the libm oracle, TRG_BVH_QUADS on (the default)."""
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from toyraygun_amd import capi
from oracle import pyoracle as O


def edge_length_per_pixel(positions, w, h):
    """Sum over all triangle edges of their length on the screen in pixels (clipped to the image rectangle; edges with an end behind the camera are
    clipped at w = 1e-3), divided by the number of pixels."""
    u = O.make_uniforms(w, h)
    inv = np.array(list(u.inv_view_proj), np.float64).reshape(4, 4).T      # world = clip . inv  (Raytracing.metal:60-85 as trg_device.h raygen evaluates it)
    vp = np.linalg.inv(inv)
    P = np.asarray(positions, np.float64).reshape(-1, 3, 3)
    clip = np.concatenate([P, np.ones(P.shape[:2] + (1,))], -1) @ vp          # [n, 3, 4]
    total = 0.0
    for a, b in ((0, 1), (1, 2), (2, 0)):
        A, B = clip[:, a].copy(), clip[:, b].copy()
        # clip against w >= eps
        eps = 1e-3
        wa, wb = A[:, 3], B[:, 3]
        both_behind = (wa < eps) & (wb < eps)
        ta = np.where(wa < eps, (eps - wa) / np.where(wb - wa == 0, 1, wb - wa), 0.0)
        tb = np.where(wb < eps, (eps - wb) / np.where(wa - wb == 0, 1, wa - wb), 0.0)
        A2 = A + (B - A) * ta[:, None]
        B2 = B + (A - B) * tb[:, None]
        pa = (A2[:, :2] / A2[:, 3:4] * 0.5 + 0.5) * (w, h)
        pb = (B2[:, :2] / B2[:, 3:4] * 0.5 + 0.5) * (w, h)
        # Liang-Barsky against [0, w] x [0, h]
        d = pb - pa
        t0, t1 = np.zeros(len(pa)), np.ones(len(pa))
        ok = ~both_behind
        for k, (lo, hi) in enumerate(((0.0, float(w)), (0.0, float(h)))):
            for sgn, bound in ((-1.0, lo), (1.0, hi)):
                p = sgn * d[:, k]
                q = sgn * (bound - pa[:, k]) if sgn > 0 else (pa[:, k] - bound)
                with np.errstate(divide="ignore", invalid="ignore"):
                    r = q / p
                ok &= ~((p == 0) & (q < 0))
                t0 = np.where(p < 0, np.maximum(t0, r), t0)
                t1 = np.where(p > 0, np.minimum(t1, r), t1)
        ok &= t0 < t1
        total += float((np.linalg.norm(d, axis=1) * np.clip(t1 - t0, 0, 1))[ok].sum())
    return total / (w * h)


def per_ray(cases, seed):
    """The fuzz regime (scripts/gpu_fuzz.py: images up to 90 x 70, 1 - 40 spp, 0 - 6 bounces, soups of up to 6,000 triangles): outlier pixels per
    MILLION RAYS TRACED, by triangle count -- what the fuzz's bar is made of (tests/util.py edge_flip_allowance)."""
    rng = np.random.default_rng(seed)
    eye = np.eye(4, dtype=np.float32)
    buckets = {}
    t_start = time.time()
    for case in range(cases):
        n = int(rng.choice([0, 100, 400, 1000, 2000, 4000, 6000, 9000]))
        s = O.OracleScene.cornell_box()
        if n:
            ctr = rng.uniform([-0.9, 0.1, -0.9], [0.9, 1.9, 0.9], (n, 3)).astype(np.float32)
            tri = ctr[:, None, :] + rng.normal(0, rng.choice([0.02, 0.1, 0.4]), (n, 3, 3)).astype(np.float32)
            if n > 10:
                tri[n // 2: n // 2 + n // 10] = tri[: n // 10]      # duplicates, as the fuzz has them
            for k in range(n):
                s.add_geometry(tri[k], [0, 1, 2], eye, rng.uniform(0.2, 0.9, 3), 1)
        b = s.buffers()
        w, h = int(rng.integers(32, 90)), int(rng.integers(32, 70))
        spp, bnc = int(rng.integers(1, 41)), int(rng.integers(1, 7))
        off = O.pixel_offsets(w, h, seed=int(rng.integers(1, 2 ** 31)))
        ref, rst = O.render(s, w, h, spp, bnc, offsets=off)
        c = capi.Context(w, h)
        try:
            c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
            c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
            c.set_pixel_offsets(off)
            c.render(0, spp, bnc)
            img = c.read_accum()
        finally:
            c.close()
        dd = np.linalg.norm(img[..., :3].astype(np.float64) - ref[..., :3], axis=-1)
        nr = np.linalg.norm(ref[..., :3].astype(np.float64), axis=-1)
        out = int((dd > 1e-4 * np.maximum(1.0, nr)).sum())
        bk = buckets.setdefault(n + 36, [0, 0, 0, 0, 0.0])
        bk[0] += 1; bk[1] += out; bk[2] += int(rst.rays); bk[3] = max(bk[3], out); bk[4] = max(bk[4], out / (w * h))
    print("fuzz regime, %d cases, %.0f s" % (cases, time.time() - t_start))
    for n in sorted(buckets):
        k, out, rays, worst, worst_share = buckets[n]
        print("  %5d triangles: %3d cases, %4d outlier pixels in %7.1f M rays = %.2f per million rays; worst image %d pixels (share %.4f)" % (n, k, out, rays / 1e6, out / max(rays, 1) * 1e6, worst, worst_share), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "per_ray":
        return per_ray(int(sys.argv[3]) if len(sys.argv) > 3 else 400, int(sys.argv[1]))
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
    eye = np.eye(4, dtype=np.float32)
    rows = []
    t_start = time.time()
    for n in (0, 30, 300, 1000, 3000, 6000, 10000):
        for res in (32, 64, 128, 256):
            for rep in range(3 if n else 1):
                s = O.OracleScene.cornell_box()
                if n:
                    ctr = rng.uniform([-0.9, 0.1, -0.9], [0.9, 1.9, 0.9], (n, 3)).astype(np.float32)
                    tri = ctr[:, None, :] + rng.normal(0, rng.choice([0.02, 0.1, 0.4]), (n, 3, 3)).astype(np.float32)
                    for k in range(n):
                        s.add_geometry(tri[k], [0, 1, 2], eye, rng.uniform(0.2, 0.9, 3), 1)
                b = s.buffers()
                w, h = res, int(res * rng.choice([0.75, 1.0]))
                spp, bnc = int(rng.choice([1, 2, 4, 8, 16, 32])), int(rng.integers(1, 7))
                off = O.pixel_offsets(w, h, seed=int(rng.integers(1, 2 ** 31)))
                ref, _ = O.render(s, w, h, spp, bnc, offsets=off)
                c = capi.Context(w, h)
                try:
                    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
                    c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h)))
                    c.set_pixel_offsets(off)
                    c.render(0, spp, bnc)
                    img = c.read_accum()
                finally:
                    c.close()
                dd = np.linalg.norm(img[..., :3].astype(np.float64) - ref[..., :3], axis=-1)
                nr = np.linalg.norm(ref[..., :3].astype(np.float64), axis=-1)
                out = int((dd > 1e-4 * np.maximum(1.0, nr)).sum())
                lam = edge_length_per_pixel(b["positions"], w, h)
                rows.append((n + 36, w, h, spp, bnc, lam, out, out / (w * h)))
                print("tris %5d %3dx%-3d spp %2d bounces %d  lambda %7.2f  outliers %4d  share %.5f  share/lambda %.2e" % (n + 36, w, h, spp, bnc, lam, out, out / (w * h), out / (w * h) / max(lam, 1e-9)), flush=True)
    r = np.array(rows)
    ratio = r[:, 7] / np.maximum(r[:, 5], 1e-9)
    print("cases %d, %.0f s; share / lambda: median %.2e, p95 %.2e, max %.2e  (lambda range %.2f .. %.1f)" % (len(r), time.time() - t_start, np.median(ratio), np.percentile(ratio, 95), ratio.max(), r[:, 5].min(), r[:, 5].max()))
    big = r[r[:, 5] >= 1.0]
    if len(big):
        rb = big[:, 7] / big[:, 5]
        print("cases with lambda >= 1: share / lambda median %.2e, max %.2e" % (np.median(rb), rb.max()))


if __name__ == "__main__":
    main()
