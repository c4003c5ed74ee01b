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


def test_c_abi_row_band_layout_rehearsal(O):
    """The multi-GPU path of the C ABI (trg_group_*, toyraygun_amd/csrc/trg_group.cpp) shards by bands of B = ceil(h / G) rows and
    gathers IN PLACE: device g's band sits at float offset g * B * w * 4 of a frame buffer padded to G * B rows, which is the
    layout ncclAllGather defines as in-place (sendbuff = recvbuff + rank * count).  Rehearsed here without GPUs: trg_band_rows
    (host-only) gives the bands, the oracle renders each device's band, the exchange is replayed with numpy exactly as
    trg_group_render enqueues it, and the first h rows of every device's buffer must equal the unsharded frame bit for bit --
    for heights that divide evenly, that do not, and for more devices than rows."""
    from toyraygun_amd import capi
    capi.load()
    scene = O.OracleScene.cornell_box()
    for (w, h, G) in ((40, 32, 2), (40, 30, 4), (24, 17, 3), (16, 5, 8), (32, 24, 1)):
        bands = [capi.band_rows(h, G, r) for r in range(G)]
        B = -(-h // G)
        # the bands tile [0, h) in order, every band has B rows except the tail
        assert bands[0][0] == 0 and sum(n for _, n in bands) == h
        for r in range(G):
            assert bands[r][0] == min(h, r * B) and bands[r][1] == max(0, min(h, (r + 1) * B) - r * B)
        off = O.pixel_offsets(w, h)
        full, fst = O.render(scene, w, h, 2, 2, offsets=off)
        count = B * w * 4
        frames, rays = [], 0
        for r in range(G):
            buf = np.zeros((G * B, w, 4), np.float32)            # the padded frame buffer of device r
            row0, rows = bands[r]
            if rows:
                acc = np.zeros((h, w, 4), np.float32)
                _, st = O.render(scene, w, h, 2, 2, row0=row0, rows=rows, accum=acc, offsets=off)
                buf[row0:row0 + rows] = acc[row0:row0 + rows]
                rays += st.rays
            frames.append(buf)
        flat = [f.reshape(-1) for f in frames]
        # TRG_GATHER_ALL: rank r's `count` floats at offset r * count land at the same offset of every rank
        gathered = [f.copy() for f in flat]
        for dst in range(G):
            for src in range(G):
                gathered[dst][src * count:(src + 1) * count] = flat[src][src * count:(src + 1) * count]
        for dst in range(G):
            got = gathered[dst].reshape(G * B, w, 4)[:h]
            assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), (w, h, G, dst)
        # TRG_GATHER_ROOT: only the root receives
        root = G - 1
        rootbuf = flat[root].copy()
        for src in range(G):
            if src != root:
                rootbuf[src * count:(src + 1) * count] = flat[src][src * count:(src + 1) * count]
        assert np.array_equal(rootbuf.reshape(G * B, w, 4)[:h].view(np.uint32), full.view(np.uint32))
        assert rays == fst.rays


def test_microband_rule_and_unpack_layout():
    """INTERLEAVED bands (trg_render_bands / trg_unpack_bands, round 4): dist.microband_rows is trg_microband_rows; the ranks' micro-bands
    partition the image rows; trg_unpack_bands' index map (dist.unpack_bands_reference) is the inverse of the compact placement."""
    from toyraygun_amd import capi
    from toyraygun_amd.dist import MICRO_BAND_ROWS, microband_rows, unpack_bands_reference
    capi.load()
    for h in (1, 5, 8, 9, 93, 135, 1080, 2160):
        for world in (1, 2, 3, 4, 8, 11):
            rule = [microband_rows(h, world, r) for r in range(world)]
            assert rule == [capi.microband_rows(h, world, r) for r in range(world)], (h, world)
            stride = rule[0][1]
            assert all(s == stride for _, s in rule) and stride % MICRO_BAND_ROWS == 0 and all(n <= stride for n, _ in rule)
            nmb = -(-h // MICRO_BAND_ROWS)
            assert sum(n for n, _ in rule) == nmb * MICRO_BAND_ROWS           # every micro-band has exactly one owner
            # place image row y where its owner stores it, then unpack: the identity
            img = np.arange(h, dtype=np.int64)
            compact = np.full(world * stride, -1, np.int64)
            for y in range(h):
                mb = y // MICRO_BAND_ROWS
                r, l = mb % world, (mb // world) * MICRO_BAND_ROWS + y % MICRO_BAND_ROWS
                assert l < rule[r][0]
                assert compact[r * stride + l] == -1
                compact[r * stride + l] = y
            assert np.array_equal(unpack_bands_reference(compact, h, world), img)
    assert microband_rows(1080, 8, 0) == (136, 136) and microband_rows(1080, 8, 7) == (128, 136) and microband_rows(93, 3, 2) == (32, 32)


def _interleaved_worker(rank, world, port, h, w, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    from toyraygun_amd.dist import MICRO_BAND_ROWS, gather_bands, microband_rows, unpack_bands_reference
    scene = O.OracleScene.cornell_box()
    rows, stride = microband_rows(h, world, rank)
    compact = np.zeros((world * stride, w, 4), np.float32)
    acc = np.zeros((h, w, 4), np.float32)
    nmb = -(-h // MICRO_BAND_ROWS)
    for k, mb in enumerate(range(rank, nmb, world)):          # the oracle renders this rank's micro-bands; stored as trg_render_bands stores them
        y0, n = mb * MICRO_BAND_ROWS, min(MICRO_BAND_ROWS, h - mb * MICRO_BAND_ROWS)
        O.render(scene, w, h, 2, 3, row0=y0, rows=n, accum=acc, nthreads=2)
        compact[rank * stride + k * MICRO_BAND_ROWS: rank * stride + k * MICRO_BAND_ROWS + n] = acc[y0:y0 + n]
    full = torch.from_numpy(compact)
    gather_bands(full, world, rank)                            # equal slices of `stride` rows: the in-place all-gather
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), unpack_bands_reference(full.numpy(), h, world))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 32), (3, 37)])
def test_interleaved_bands_equal_single(tmp_path, world, h, O):
    """The launched multi-rank path with interleaved micro-bands, rehearsed over gloo with the oracle as the per-rank renderer: compact
    bands, ONE in-place all-gather of equal slices, the unpack -- every rank ends with the unsharded frame bit for bit."""
    w = 40
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_interleaved_worker, args=(world, port, h, w, str(tmp_path)), nprocs=world, join=True)
    ref, _ = O.render(O.OracleScene.cornell_box(), w, h, 2, 3)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, ref), "rank %d frame differs from the unsharded render" % r
