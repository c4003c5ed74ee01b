"""Scenes of the measurement scripts beyond the BASELINE configurations (scripts/exp_ab.py, scripts/c4_run.py)."""


def icosphere(k):
    """(positions [n, 3], triangles [m, 3]) of a unit icosphere after k subdivisions (m = 20 * 4^k), numpy only."""
    import numpy as np
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(k):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.minimum(e[:, 0], e[:, 1]) * len(v) + np.maximum(e[:, 0], e[:, 1])
        uniq, inv = np.unique(key, return_inverse=True)
        mid = v[uniq // len(v)] + v[uniq % len(v)]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = inv.reshape(3, -1).T + len(v)      # midpoints of edges (01, 12, 20) of every face
        v = np.concatenate([v, mid])
        f = np.concatenate([np.stack([f[:, 0], m[:, 0], m[:, 2]], 1), np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
                            np.stack([f[:, 2], m[:, 2], m[:, 1]], 1), m])
    return v.astype(np.float32), f.astype(np.uint32)


def make_scene(host, sc):
    if sc.startswith("lattice"):
        return host.Scene.cornell_lattice(int(sc[7:] or 44))
    s = host.Scene.cornell_box()
    if sc.startswith("sphere"):
        v, f = icosphere(int(sc[6:]))
        s.add_mesh(v, v, f, host.mtx_srt((0.3, 0.3, 0.3), (0.0, 0.0, 0.0), (0.0, 1.45, 0.15)), (0.75, 0.75, 0.75), 1)   # material 1 = TRG_MATERIAL_DEFAULT
    return s
