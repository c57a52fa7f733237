"""How one frame (or one accumulated frame) is split over the GPUs of a node, one process per GPU.

The reference's only parallelism is 200 independent pixel chunks per frame (src/renderer.rs:104-136); pixels and
samples are independent, so the work shards without any exchange inside the per-pixel path:

* rows     each rank renders a contiguous band of rows of the same frame; the bands are disjoint ranges of the frame
           buffer, so the only collective is an all-gather of RGBA8 bands when one rank needs the whole frame.
* samples  each rank renders a subset of the sample indices of an accumulated frame into its own accumulator
           (Renderer.accumulated, src/renderer.rs:14,59-66); one reduce-scatter(sum) of the f32 accumulators leaves
           every rank with the summed rows it owns (half the traffic of an all-reduce), it resolves them (divide by
           the sample count, clamp, pack), and the RGBA8 bands are all-gathered (1/4 of the accumulator's bytes).
* frames   whole frames of a sequence go to different ranks; no collective.

The functions take a `render` callable so that the same logic drives the HIP backend (bench.py) and, in the CPU tests,
the oracle under the gloo backend.
"""
from __future__ import annotations

from typing import Callable, List, Tuple


def shard_rows(height: int, rank: int, world_size: int, band: int = 8) -> Tuple[int, int]:
    """Row band [begin, end) of `rank`: contiguous, aligned to the kernels' 8-row tiles, sizes differ by at most one tile."""
    n_bands = (height + band - 1) // band
    base, extra = divmod(n_bands, world_size)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return min(first * band, height), min((first + count) * band, height)


def shard_samples(n_samples: int, rank: int, world_size: int, first_index: int = 1) -> List[int]:
    """Sample (frame) indices rendered by `rank`: first_index + rank, + world_size, ... (interleaved)."""
    return list(range(first_index + rank, first_index + n_samples, world_size))


def gather_row_bands(dist, local_rgba, height: int, rank: int, world_size: int):
    """All-gather of RGBA8 row bands into the full frame on every rank.

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


def equal_row_bands(height: int, world_size: int) -> bool:
    """True when the frame splits into `world_size` bands of the same number of rows (what reduce-scatter needs)."""
    return world_size >= 1 and height % world_size == 0


def reduce_scatter_accumulators(dist, accum, rank: int, world_size: int):
    """Sums the (h, w, 4) f32 accumulators of all ranks; returns (row_begin, row_end, band) where `band` holds the
    summed rows [row_begin, row_end) of this rank.  Equal bands go through one reduce-scatter; otherwise (or when the
    backend lacks it) an all-reduce followed by a slice, with the 8-row-aligned bands of shard_rows."""
    import torch
    h = int(accum.shape[0])
    if world_size == 1 or dist is None:
        return 0, h, accum
    if equal_row_bands(h, world_size):
        rows = h // world_size
        band = torch.empty((rows,) + tuple(accum.shape[1:]), dtype=accum.dtype, device=accum.device)
        try:
            dist.reduce_scatter_tensor(band, accum)
            return rank * rows, (rank + 1) * rows, band
        except (RuntimeError, NotImplementedError):
            pass
    dist.all_reduce(accum)
    b, e = shard_rows(h, rank, world_size)
    return b, e, accum[b:e]


def gather_equal_bands(dist, band_rgba, full_rgba) -> None:
    """All-gather of equal RGBA8 row bands (rank order = row order) into `full_rgba` on every rank."""
    dist.all_gather_into_tensor(full_rgba, band_rgba)
