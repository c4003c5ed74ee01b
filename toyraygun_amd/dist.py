"""Multi-GPU rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Pixels are independent, so the frame is split into contiguous ROW BANDS: rank g renders rows
[g*B, min(h, (g+1)*B)), B = ceil(h / G) (band_rows: the rule of trg_band_rows), for all its samples with no communication, writing straight into its slice of a
full-frame device tensor; ONE all-gather at the end lands every band in place on every rank (disjoint
pixels: no reduction, no change of floating-point summation order).  The reference has no multi-GPU
code at all; this is new.

Streams: the megakernel runs on render streams, the collective on a communication stream, ordered by
events.  With `pipelined=True` every frame has its own buffer, so the gather of frame k overlaps the
renders that follow, and consecutive frames render on FOUR alternating streams: a launch ends with a
tail in which most CUs are idle behind the last workgroups (a workgroup of a full C2 frame lives about
a quarter of the launch) and a small row band never fills the chip at all; the next frames' workgroups
use those CUs instead of waiting for the drain.  Frames are independent images (own buffer, own
stream); within a frame nothing changes, and the library is told how many launches overlap
(TRG_OPT_LAUNCHES_IN_FLIGHT: separate stack scratch per launch, frame split chosen accordingly).
"""
import os

# launches overlap only when their streams sit on different hardware queues; HIP spreads streams round-robin over
# GPU_MAX_HW_QUEUES (default 4).  Read when the HIP runtime starts, so this only helps if nothing touched the GPU yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


_INPLACE_OK = True


def band_rows(height, world, rank):
    """Rows [row0, row0+rows) of rank `rank`: bands of ceil(height / world) rows, the last one(s) shorter or empty -- the ONE band
    rule of the project (trg_band_rows in include/trg.h does the same arithmetic; tests assert they agree)."""
    band = -(-height // world) if world else height
    row0, row1 = min(height, band * rank), min(height, band * (rank + 1))
    return row0, row1 - row0


MICRO_BAND_ROWS = 8   # trg_kernels.h kMicroBandRows: the height of a wavefront's sub-tile


def microband_rows(height, world, rank):
    """(rows of rank's INTERLEAVED band, the ranks' common stride in rows): rank renders the 8-row micro-bands rank, rank + world, ... of the
    image and stores them compactly in rows [rank * stride, rank * stride + rows) of a (world * stride)-row buffer -- the rule of
    trg_microband_rows (include/trg.h; tests assert they agree)."""
    nmb = -(-height // MICRO_BAND_ROWS)
    mine = -(-(nmb - rank) // world) if nmb > rank else 0
    return mine * MICRO_BAND_ROWS, -(-nmb // world) * MICRO_BAND_ROWS


def unpack_bands_reference(compact, height, world):
    """What trg_unpack_bands computes, in numpy / torch indexing (tests, CPU rehearsals): image row y <- compact row r * stride + l with
    r = (y // 8) % world, l = (y // 8 // world) * 8 + y % 8."""
    _, stride = microband_rows(height, world, 0)
    ys = list(range(height))
    src = [((y // MICRO_BAND_ROWS) % world) * stride + (y // MICRO_BAND_ROWS // world) * MICRO_BAND_ROWS + y % MICRO_BAND_ROWS for y in ys]
    return compact[src]


def gather_bands(full, world, rank, group=None, root=None, mode=None):
    """Exchange the row bands of `full` ([h, w, 4] float32, this rank's band already filled) in place.
    root = r: GATHER to rank r only (the other ranks keep just their own band) -- north_star's exchange: the root ingests world - 1 bands over
    world - 1 independent xGMI links, one eighth of the all-gather's traffic at world = 8; also selected by TRG_GATHER=root (rank 0).
    root = None: all-gather, every rank ends with the whole frame.  mode = "all" / "root" states the choice outright (the environment
    variable is then not consulted)."""
    if world == 1 and not os.environ.get("TRG_FORCE_GATHER"):
        return full
    if mode == "all":
        root = None
    elif mode == "root":
        root = 0 if root is None else root
    elif root is None and os.environ.get("TRG_GATHER") == "root":
        root = 0
    if full.is_cuda and dist.get_backend(group) == "gloo":
        # no RCCL in this process group (a machine without it, or the one-GPU rehearsal of bench.py's launched path, where the ranks share a
        # device and NCCL refuses a communicator): the bands go through host memory -- correct, slow, never what a scaling number is taken with
        host = full.cpu()   # (waits for the current stream, which is ordered behind the render)
        gather_bands(host, world, rank, group, root, mode)
        full.copy_(host)
        return full
    if root is not None:
        h = full.shape[0]
        bands = [band_rows(h, world, r) for r in range(world)]
        mine = full[bands[rank][0]:bands[rank][0] + bands[rank][1]]
        if len({b[1] for b in bands}) == 1:
            # the receive list ARE the row bands of the frame: every band lands in place
            dist.gather(mine, [full[b0:b0 + n] for b0, n in bands] if rank == root else None, dst=root, group=group)
        else:
            rows_max = max(b[1] for b in bands)
            pad = torch.zeros((rows_max,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
            pad[:bands[rank][1]] = mine
            out = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
            dist.gather(pad, out, dst=root, group=group)
            if rank == root:
                for r, (r0, n) in enumerate(bands):
                    if r != rank:
                        full[r0:r0 + n] = out[r][:n]
        return full
    h = full.shape[0]
    bands = [band_rows(h, world, r) for r in range(world)]
    equal = len({b[1] for b in bands}) == 1
    mine = full[bands[rank][0]:bands[rank][0] + bands[rank][1]]
    if equal and hasattr(dist, "all_gather_into_tensor"):
        # in place: the output IS the frame, the input is this rank's slice of it (NCCL's in-place all-gather layout:
        # sendbuff == recvbuff + rank * count).  Should a backend refuse the aliasing, send a copy of the band instead.
        global _INPLACE_OK
        if _INPLACE_OK:
            try:
                dist.all_gather_into_tensor(full, mine, group=group)
                return full
            except (RuntimeError, ValueError):
                _INPLACE_OK = False
        dist.all_gather_into_tensor(full, mine.clone(), group=group)
        return full
    rows_max = max(b[1] for b in bands)
    pad = torch.zeros((rows_max,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    pad[:bands[rank][1]] = mine
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    for r, (r0, n) in enumerate(bands):
        if r != rank:
            full[r0:r0 + n] = out[r][:n]
    return full


class DistributedRenderer:
    """Row-band sharded renderer over an initialised process group (one rank per GPU)."""

    def __init__(self, width, height, device_index, group=None, pipelined=False, depth=None, interleaved=None, gather_mode=None):
        from . import capi
        self.capi = capi
        self.group = group
        # what render(gather=True) does: "root" = the gather to rank 0 north_star names (default; the other ranks keep their own band),
        # "all" = the in-place all-gather (every rank ends with the frame).  TRG_GATHER=all|root overrides the default.
        self.gather_mode = gather_mode or os.environ.get("TRG_GATHER") or "root"
        if self.gather_mode not in ("root", "all"):
            raise ValueError("gather mode must be 'root' or 'all' (TRG_GATHER), not %r" % (self.gather_mode,))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.w, self.h = width, height
        self.device = torch.device("cuda", device_index)
        self.row0, self.rows = band_rows(height, self.world, self.rank)
        # INTERLEAVED bands (trg_render_bands): 8-row micro-bands dealt round robin even out what the bands cost (contiguous bands of C2 at 8
        # GPUs: slowest / mean 1.05, of the million-triangle scene 1.2).  The band is rendered compactly into this rank's slice of a
        # (world * stride)-row buffer, the in-place all-gather completes the compact frame, trg_unpack_bands turns it into the image on
        # the communication stream.  Default: on for more than one rank (TRG_BANDS=contiguous turns it off).
        if interleaved is None:
            interleaved = self.world > 1 and os.environ.get("TRG_BANDS", "interleaved") != "contiguous"
        self.interleaved = bool(interleaved)
        self.il_rows, self.il_stride = microband_rows(height, self.world, self.rank)
        if depth is None:
            # frames in flight: 2 hide the tail of a launch that fills the chip; a small band (a fraction of one resident set
            # of workgroups) needs more to fill it at all.  4 is never worse than 2 (C2 ms per step, 2 -> 4 in flight: full
            # frame 2.05 -> 2.05, 1/4 0.52 -> 0.50, 1/8 0.31 -> 0.27), costs three more 33 MB buffers and, with the
            # communication stream, the context's and torch's, still fits the 8 hardware queues bench.py asks for.
            depth = int(os.environ.get("TRG_PIPE_DEPTH", "4"))   # (the environment variable is for measurements)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(width, height, device=device_index)
        # torch owns the frames (so RCCL can see them); the kernel writes into them through trg_bind_accum
        nslots = depth if pipelined else 1
        if self.interleaved:
            self.compact = [torch.zeros((self.world * self.il_stride, width, 4), dtype=torch.float32, device=self.device) for _ in range(nslots)]
        self.frames = [torch.zeros((height, width, 4), dtype=torch.float32, device=self.device) for _ in range(nslots)]
        self.render_streams = [torch.cuda.Stream(self.device) for _ in range(depth if pipelined else 1)]
        self.render_stream = self.render_streams[0]
        self.comm_stream = torch.cuda.Stream(self.device) if pipelined else self.render_stream
        self.ctx.set_stream(self.render_stream.cuda_stream)
        self.ctx.bind_accum((self.compact if self.interleaved else self.frames)[0].data_ptr())
        self._step = 0
        self._gathered = [None] * len(self.frames)  # event: last gather into frames[i] has finished
        self.frame = self.frames[0]
        self._needs_gather = self.world > 1 or bool(os.environ.get("TRG_FORCE_GATHER"))
        self._last_stream = self.render_stream
        self._overlap = False
        self.time_launches = False   # bracket every launch with timing events on its stream (no host sync); see launch_ms()
        self._timed = []

    def load_scene(self, buffers):
        self.ctx.load_scene(buffers["positions"], buffers["normals"], buffers["colors"], buffers["indices"], buffers["material_ids"])
        # tell the library how many launches overlap: each gets its own traversal-stack scratch, and the frame split adapts
        self._overlap = len(self.render_streams) > 1
        self.ctx.set_option(self.capi.OPT_LAUNCHES_IN_FLIGHT, len(self.render_streams) if self._overlap else 1)
        if self._overlap:
            # trg_render's default timing brackets every launch with events and WAITS for the second one (a host sync
            # per launch): nothing would ever be in flight.  Pipelined rendering is asynchronous; per-launch times come
            # from events on the launch's own stream (time_launches / launch_ms()).
            self.ctx.set_option(self.capi.OPT_TIMING, 0)

    def render(self, frame_begin, spp, bounces, gather=True):
        """Render this rank's band and exchange: gather=True -> self.gather_mode ("root": rank 0 ends with the whole frame, the others with
        their own band; "all": every rank with the whole frame), or "root" / "all" outright; False: no exchange.
        Returns the frame tensor the result lands in (valid after synchronize())."""
        mode = self.gather_mode if gather is True else gather
        if mode not in (False, None, "root", "all"):
            raise ValueError("gather must be True, False, 'root' or 'all'")
        i = self._step % len(self.frames)
        self._step += 1
        frame = self.frames[i]
        target = self.compact[i] if self.interleaved else frame      # what the kernel writes: the compact band buffer, or the image itself
        rs = self.render_streams[i] if getattr(self, "_overlap", False) else self.render_stream
        if len(self.frames) > 1:
            self.ctx.bind_accum(target.data_ptr())
            if rs is not self._last_stream:
                self.ctx.set_stream(rs.cuda_stream)
                self._last_stream = rs
            if self._gathered[i] is not None:  # the previous gather into this buffer must be done
                rs.wait_event(self._gathered[i])
        if self.time_launches:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(rs)
            self._launch(frame_begin, spp, bounces)
            e1.record(rs)
            self._timed.append((e0, e1))
        else:
            self._launch(frame_begin, spp, bounces)
        gathered = bool(mode) and self._needs_gather
        if gathered:
            if self.comm_stream is not rs:
                done = torch.cuda.Event()
                done.record(rs)
                self.comm_stream.wait_event(done)
            with torch.cuda.stream(self.comm_stream):
                gather_bands(target, self.world, self.rank, self.group, mode=mode)   # (a compact frame always has equal bands: in place)
                if self.interleaved:
                    self._unpack(target, frame, self.comm_stream, rs)
                if len(self.frames) > 1:
                    ev = torch.cuda.Event()
                    ev.record(self.comm_stream)
                    self._gathered[i] = ev
        elif self.interleaved:
            self._unpack(target, frame, rs, rs)       # this rank's micro-bands alone, right behind the render
        self.frame = frame
        return frame

    @property
    def owned_rows(self):
        """Image rows this rank renders (its contiguous band, or the rows of its micro-bands that lie inside the image)."""
        if not self.interleaved:
            return self.rows
        nmb = -(-self.h // MICRO_BAND_ROWS)
        return sum(min(MICRO_BAND_ROWS, self.h - j * MICRO_BAND_ROWS) for j in range(self.rank, nmb, self.world))

    def launch_band(self, frame_begin, spp, bounces):
        """This rank's band into the currently bound buffer on the context's current stream, nothing else (counters / launch-alone passes)."""
        self._launch(frame_begin, spp, bounces)

    def _launch(self, frame_begin, spp, bounces):
        if self.interleaved:
            self.ctx.render_bands(frame_begin, spp, bounces, self.world, self.rank, self.rank * self.il_stride)
        else:
            self.ctx.render(frame_begin, spp, bounces, self.row0, self.rows)

    def _unpack(self, compact, frame, on, back_to):
        """trg_unpack_bands on stream `on` (the context launches on its current stream: switched for the call, a host-side pointer)."""
        if on is not back_to:
            self.ctx.set_stream(on.cuda_stream)
        self.ctx.unpack_bands(compact.data_ptr(), frame.data_ptr(), self.world)
        if on is not back_to:
            self.ctx.set_stream(back_to.cuda_stream)

    def launch_ms(self):
        """Durations (ms) of the launches rendered while time_launches was set, from events on their own streams; call after
        synchronize().  Overlapping launches share the GPU, so each lasts longer than it would alone."""
        out = [a.elapsed_time(b) for a, b in self._timed]
        self._timed = []
        return out

    def synchronize(self):
        for rs in self.render_streams:
            rs.synchronize()
        self.comm_stream.synchronize()

    def close(self):
        self.synchronize()
        self.ctx.close()
