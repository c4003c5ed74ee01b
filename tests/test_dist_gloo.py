"""World-size-2 (and 3, uneven bands) CPU rehearsal of the multi-GPU path: row-band sharding + one all-gather,
with the oracle standing in for the GPU renderer on each rank (gloo backend)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_rows_partition():
    from toyraygun_amd.dist import band_rows
    for h in (1, 7, 64, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            rows = [band_rows(h, world, r) for r in range(world)]
            assert rows[0][0] == 0 and sum(n for _, n in rows) == h
            for (a, n), (b, _) in zip(rows, rows[1:]):
                assert a + n == b
    assert band_rows(1080, 8, 3) == (405, 135) and band_rows(2160, 8, 7) == (1890, 270)


def _worker(rank, world, port, h, w, out_dir, root=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    from toyraygun_amd.dist import band_rows, gather_bands
    scene = O.OracleScene.cornell_box()
    row0, rows = band_rows(h, world, rank)
    acc = np.zeros((h, w, 4), np.float32)
    O.render(scene, w, h, 2, 3, row0=row0, rows=rows, accum=acc, nthreads=2)
    full = torch.from_numpy(acc)
    gather_bands(full, world, rank, root=root)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 32), (3, 34)])
def test_sharded_render_equals_single(tmp_path, world, h, O):
    w = 48
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, h, w, str(tmp_path)), nprocs=world, join=True)
    ref, _ = O.render(O.OracleScene.cornell_box(), w, h, 2, 3)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, ref), "rank %d frame differs from the unsharded render" % r


@pytest.mark.parametrize("world,h", [(2, 32), (3, 34)])
def test_gather_to_root_only(tmp_path, world, h, O):
    """gather_bands(root=0): rank 0 ends up with the unsharded frame bit for bit; the others keep their own band only."""
    from toyraygun_amd.dist import band_rows
    w = 48
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, h, w, str(tmp_path), 0), nprocs=world, join=True)
    ref, _ = O.render(O.OracleScene.cornell_box(), w, h, 2, 3)
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "rank0.npy")), ref)
    for r in range(1, world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        r0, n = band_rows(h, world, r)
        assert np.array_equal(got[r0:r0 + n], ref[r0:r0 + n])
        assert (np.delete(got, np.s_[r0:r0 + n], axis=0) == 0).all()
