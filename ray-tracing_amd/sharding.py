"""How one frame (or one accumulated frame) is split over the GPUs of a node, one process per GPU.

The reference's only parallelism is 200 independent pixel chunks per frame (src/renderer.rs:104-136); pixels and
samples are independent, so the work shards without any exchange inside the per-pixel path:

* rows     each rank renders every world_size-th 8-row tile row of the same frame (interleaved, so that sky and ground
           rows are dealt evenly: contiguous bands differ several-fold in cost); one all-gather of RGBA8 tile rows leaves
           the whole frame on every rank.
* samples  each rank renders a subset of the sample indices of an accumulated frame into its own accumulator
           (Renderer.accumulated, src/renderer.rs:14,59-66); one reduce-scatter(sum) of the f32 accumulators leaves
           every rank with the summed rows it owns (half the traffic of an all-reduce), it resolves them (divide by
           the sample count, clamp, pack), and the RGBA8 bands are all-gathered (1/4 of the accumulator's bytes).
* frames   whole frames of a sequence go to different ranks; no collective.

`FrameExchange` owns every buffer the collectives touch, allocated once, so that a step allocates nothing.  The same
class drives the HIP backend (bench.py, device tensors, nccl = RCCL) and, in the CPU tests, the oracle under gloo.
Which collective runs is decided up front from the backend name and the frame height, never by catching an error:
a failing collective propagates.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

TILE_ROWS = 8  # rows of one wavefront tile of the render kernels: the granularity of row sharding


def shard_rows(height: int, rank: int, world_size: int, band: int = TILE_ROWS) -> Tuple[int, int]:
    """Contiguous row band [begin, end) of `rank`, aligned to the kernels' 8-row tiles, sizes differ by at most one tile."""
    n_bands = (height + band - 1) // band
    base, extra = divmod(n_bands, world_size)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return min(first * band, height), min((first + count) * band, height)


def interleaved_tile_rows(height: int, rank: int, world_size: int) -> List[Tuple[int, int]]:
    """Row ranges [begin, end) of the 8-row tile rows `rank` renders in the rows mode: tile rows rank, rank + world_size, ..."""
    n_tiles = (height + TILE_ROWS - 1) // TILE_ROWS
    return [(t * TILE_ROWS, min((t + 1) * TILE_ROWS, height)) for t in range(rank, n_tiles, world_size)]


def shard_samples(n_samples: int, rank: int, world_size: int, first_index: int = 1) -> List[int]:
    """Sample (frame) indices rendered by `rank`: a contiguous block of the n_samples indices from first_index on (the first
    n_samples % world_size ranks take one more), so that a rank renders its share with one rt_render_samples_device call."""
    q, rem = divmod(n_samples, world_size)
    begin = first_index + rank * q + min(rank, rem)
    return list(range(begin, begin + q + (1 if rank < rem else 0)))


def equal_row_bands(height: int, world_size: int) -> bool:
    """True when the frame splits into `world_size` bands of the same number of rows (what reduce-scatter needs)."""
    return world_size >= 1 and height % world_size == 0


def backend_has_reduce_scatter(backend: str) -> bool:
    """reduce_scatter_tensor is used with RCCL only; other backends (gloo on the CPU) take the all-reduce path."""
    return backend == "nccl"


class FrameExchange:
    """Buffers and collectives of one rank for frames of `height` x `width`.

    frame   (padded_rows, width, 4) uint8: rows [0, height) are the RGBA8 frame (`img`); the padding rounds the tile-row
            count up to a multiple of world_size so that the interleaved all-gather is two strided copies and one collective.
    accum   (height, width, 4) float32: this rank's accumulator (Renderer.accumulated).
    """

    def __init__(self, dist, height: int, width: int, rank: int, world_size: int, device="cpu", backend: str = "gloo",
                 force_collectives: bool = False):
        """`force_collectives`: a single rank still issues every collective (each is the identity then), which rehearses
        the RCCL calls of a multi-GPU run on a one-GPU box."""
        import torch
        self.dist, self.h, self.w, self.rank, self.world = dist, int(height), int(width), int(rank), int(world_size)
        self.backend = backend
        self.n_tiles = (self.h + TILE_ROWS - 1) // TILE_ROWS
        self.tiles_per_rank = (self.n_tiles + self.world - 1) // self.world
        self.padded_rows = self.tiles_per_rank * self.world * TILE_ROWS
        self.frame = torch.zeros((self.padded_rows, self.w, 4), dtype=torch.uint8, device=device)
        self.accum = torch.zeros((self.h, self.w, 4), dtype=torch.float32, device=device)
        # a gloo group exchanges host memory: device tensors are staged through these (rehearsals on a box with fewer GPUs than ranks)
        self.staged = backend != "nccl" and self.frame.device.type != "cpu"
        xdev = "cpu" if self.staged else device
        self.collectives = dist is not None and (self.world > 1 or force_collectives)
        self.reduce_scatter = self.collectives and equal_row_bands(self.h, self.world) and backend_has_reduce_scatter(backend)
        if self.collectives:
            self.send_tiles = torch.zeros((self.tiles_per_rank, TILE_ROWS, self.w, 4), dtype=torch.uint8, device=xdev)
            # receive blocks are the senders' blocks concatenated along dim 0 (the layout every backend's all_gather_into_tensor accepts)
            self.recv_tiles = torch.zeros((self.world * self.tiles_per_rank, TILE_ROWS, self.w, 4), dtype=torch.uint8, device=xdev)
            if self.reduce_scatter:
                rows = self.h // self.world
                self.band_accum = torch.zeros((rows, self.w, 4), dtype=torch.float32, device=xdev)
                self.band_rgba = torch.zeros((rows, self.w, 4), dtype=torch.uint8, device=xdev)
            else:
                bands = [shard_rows(self.h, r, self.world) for r in range(self.world)]
                self.max_band = max(e - b for b, e in bands)
                self.send_band = torch.zeros((self.max_band, self.w, 4), dtype=torch.uint8, device=xdev)
                self.recv_bands = torch.zeros((self.world * self.max_band, self.w, 4), dtype=torch.uint8, device=xdev)
            if self.staged:
                self.accum_host = torch.zeros((self.h, self.w, 4), dtype=torch.float32, device="cpu")

    # ---- rows mode -------------------------------------------------------------------------------------------------
    def render_rows(self) -> Tuple[int, int, int]:
        """(row_begin, row_end, tile_row_stride) of this rank's launch in the rows mode (rt_render_params)."""
        return self.rank * TILE_ROWS, self.h, self.world

    def gather_interleaved(self) -> None:
        """All-gather of the interleaved tile rows: afterwards rows [0, height) of `frame` hold the whole frame on every rank."""
        if not self.collectives:
            return
        tiles = self.frame.view(self.tiles_per_rank, self.world, TILE_ROWS, self.w, 4)
        self.send_tiles.copy_(tiles[:, self.rank])
        self.dist.all_gather_into_tensor(self.recv_tiles, self.send_tiles)
        tiles.copy_(self.recv_tiles.view(self.world, self.tiles_per_rank, TILE_ROWS, self.w, 4).permute(1, 0, 2, 3, 4))

    # ---- samples mode ----------------------------------------------------------------------------------------------
    def sum_accumulators(self):
        """Sums `accum` over the ranks.  Returns (row_begin, row_end, band): `band` holds the summed rows this rank
        resolves -- its own tensor after a reduce-scatter (equal bands, RCCL), a slice of `accum` after an all-reduce."""
        if not self.collectives:
            return 0, self.h, self.accum
        if self.reduce_scatter:
            rows = self.h // self.world
            self.dist.reduce_scatter_tensor(self.band_accum, self.accum)
            return self.rank * rows, (self.rank + 1) * rows, self.band_accum
        if self.staged:
            self.accum_host.copy_(self.accum)
            self.dist.all_reduce(self.accum_host)
            self.accum.copy_(self.accum_host)
        else:
            self.dist.all_reduce(self.accum)
        b, e = shard_rows(self.h, self.rank, self.world)
        return b, e, self.accum[b:e]

    def gather_bands(self, row_begin: int, row_end: int) -> None:
        """All-gather of the resolved RGBA8 bands.  After a reduce-scatter the caller resolved into `band_rgba`; after an
        all-reduce into rows [row_begin, row_end) of `frame`.  Afterwards rows [0, height) of `frame` hold the whole frame."""
        if not self.collectives:
            return
        if self.reduce_scatter:
            self.dist.all_gather_into_tensor(self.frame[: self.h], self.band_rgba)   # rank order = row order
            return
        n = row_end - row_begin
        self.send_band[:n].copy_(self.frame[row_begin:row_end])
        self.dist.all_gather_into_tensor(self.recv_bands, self.send_band)
        for r in range(self.world):
            b, e = shard_rows(self.h, r, self.world)
            self.frame[b:e].copy_(self.recv_bands[r * self.max_band: r * self.max_band + (e - b)])


# ---- function forms (the exchange of one call, buffers allocated per call) -----------------------------------------
def gather_row_bands(dist, local_rgba, height: int, rank: int, world_size: int):
    """All-gather of contiguous RGBA8 row bands (shard_rows) into the full frame on every rank.

    local_rgba: (height, width, 4) uint8 tensor whose rows shard_rows(height, rank, world_size) are valid.
    Bands can differ by one 8-row tile, so they are padded to the largest band for the collective.
    """
    import torch
    bands = [shard_rows(height, r, world_size) for r in range(world_size)]
    max_rows = max(e - b for b, e in bands)
    b, e = bands[rank]
    send = torch.zeros((max_rows,) + tuple(local_rgba.shape[1:]), dtype=local_rgba.dtype, device=local_rgba.device)
    send[: e - b] = local_rgba[b:e]
    recv = [torch.empty_like(send) for _ in range(world_size)]
    dist.all_gather(recv, send)
    full = torch.empty_like(local_rgba)
    for r, (rb, re) in enumerate(bands):
        full[rb:re] = recv[r][: re - rb]
    return full


def render_sample_sharded(dist, render: Callable[[int], None], accum, n_samples: int, rank: int, world_size: int, first_index: int = 1) -> int:
    """Renders this rank's share of `n_samples` samples into `accum` (a zeroed (h, w, 4) f32 tensor) through
    `render(sample_index)`, then sums the accumulators of all ranks in place.  Returns the number of samples this rank rendered."""
    mine = shard_samples(n_samples, rank, world_size, first_index)
    for idx in mine:
        render(idx)
    if world_size > 1 and dist is not None:
        dist.all_reduce(accum)
    return len(mine)
