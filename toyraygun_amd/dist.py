"""Multi-GPU rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Pixels are independent, so the frame is split into contiguous ROW BANDS: rank g renders rows
[g*h/G, (g+1)*h/G) for all its samples with no communication, writing straight into its slice of a
full-frame device tensor; ONE all-gather at the end lands every band in place on every rank (disjoint
pixels: no reduction, no change of floating-point summation order).  The reference has no multi-GPU
code at all; this is new.
"""
import torch
import torch.distributed as dist


def band_rows(height, world, rank):
    """Rows [row0, row0+rows) of rank `rank` (the last bands absorb the remainder)."""
    row0 = (height * rank) // world
    row1 = (height * (rank + 1)) // world
    return row0, row1 - row0


def gather_bands(full, world, rank, group=None):
    """All-gather the row bands of `full` ([h, w, 4] float32, this rank's band already filled) in place."""
    if world == 1:
        return full
    h = full.shape[0]
    bands = [band_rows(h, world, r) for r in range(world)]
    equal = len({b[1] for b in bands}) == 1
    mine = full[bands[rank][0]:bands[rank][0] + bands[rank][1]]
    if equal and hasattr(dist, "all_gather_into_tensor"):
        # in place: the output IS the frame, the input is this rank's slice of it
        dist.all_gather_into_tensor(full, mine, group=group)
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

    def __init__(self, width, height, device_index, group=None):
        from . import capi
        self.capi = capi
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.w, self.h = width, height
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(width, height, device=device_index)
        # torch owns the frame (so RCCL can see it); the kernel writes into it through trg_bind_accum
        self.frame = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device)
        self.ctx.bind_accum(self.frame.data_ptr())
        # one torch stream carries both the megakernel and the collective, so they are ordered
        self.stream = torch.cuda.Stream(self.device)
        self.ctx.set_stream(self.stream.cuda_stream)
        self.row0, self.rows = band_rows(height, self.world, self.rank)

    def load_scene(self, buffers):
        self.ctx.load_scene(buffers["positions"], buffers["normals"], buffers["colors"], buffers["indices"], buffers["material_ids"])

    def render(self, frame_begin, spp, bounces, gather=True):
        """Render this rank's band; with gather=True every rank ends up with the whole frame."""
        self.ctx.render(frame_begin, spp, bounces, self.row0, self.rows)
        if gather and self.world > 1:
            with torch.cuda.stream(self.stream):
                gather_bands(self.frame, self.world, self.rank, self.group)
        return self.frame

    def synchronize(self):
        self.stream.synchronize()

    def close(self):
        self.ctx.close()
