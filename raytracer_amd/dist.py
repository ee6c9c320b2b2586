"""Multi-GPU rendering: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The image is cut into blocks of `block_rows` rows dealt round-robin to the ranks (interleaved so
that cheap sky rows and expensive sphere rows are spread evenly — SURVEY.md §5.7).  With
pixel-keyed jitter every pixel is independent, so the ranks exchange nothing while rendering; the
only communication is one gather of the finished tiles to rank 0 per frame (RCCL send/recv over
xGMI under torch.distributed.gather), after which rank 0 scatters the rows into place.

The renderer is injected as a callable so the tiling + gather logic can be tested on CPU with the
gloo backend (tests/test_dist_gloo.py) and driven by the HIP renderer in production (bench.py).
"""
from __future__ import annotations

import numpy as np

from .packing import rows_in_part


class RowTiling:
    """Row-set bookkeeping shared by all ranks."""

    def __init__(self, height, width, world_size, block_rows=16):
        self.height, self.width, self.world_size, self.block_rows = int(height), int(width), int(world_size), int(block_rows)
        self.rows = [rows_in_part(self.height, self.block_rows, p, self.world_size) for p in range(self.world_size)]
        self.max_rows = max(len(r) for r in self.rows)

    def n_rows(self, part):
        return len(self.rows[part])


def gather_tiles(tile, tiling: RowTiling, rank, *, dst=0, group=None, out=None, scratch=None, via_host=False):
    """Gather per-rank tiles (torch tensors [max_rows, W, C], rows beyond the rank's own count are
    padding) to `dst` and reassemble the frame [H, W, C] there.  Returns the frame on `dst`,
    None elsewhere.  `out` / `scratch` let the caller reuse buffers across frames."""
    import torch
    import torch.distributed as dist

    world = tiling.world_size
    if world == 1:
        frame = tile[: tiling.n_rows(0)]
        if out is not None:
            out.copy_(frame)
            return out
        return frame
    assert tile.shape[0] == tiling.max_rows and tile.is_contiguous()
    if via_host and tile.is_cuda:
        # rehearsal path (gloo cannot gather device tensors): stage through host memory
        host = tile.cpu()
        if rank == dst:
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.gather(host, gather_list=parts, dst=dst, group=group)
            if out is None:
                out = torch.empty((tiling.height, tiling.width) + tuple(tile.shape[2:]), dtype=tile.dtype, device=tile.device)
            for p in range(world):
                idx = torch.as_tensor(tiling.rows[p], device=tile.device, dtype=torch.long)
                out.index_copy_(0, idx, parts[p][: tiling.n_rows(p)].to(tile.device))
            return out
        dist.gather(host, gather_list=None, dst=dst, group=group)
        return None
    if rank == dst:
        if scratch is None:
            scratch = new_scratch(tile, world)
        dist.gather(tile, gather_list=scratch, dst=dst, group=group)
        return reassemble(scratch, tiling, out)
    dist.gather(tile, gather_list=None, dst=dst, group=group)
    return None


def reassemble(scratch, tiling: RowTiling, out=None):
    """The gather root's last step: the `world` received tiles (rows beyond a part's own count are padding) -> the frame
    [H, W, C].  Tiles made by new_scratch() + a tiling prepared by cache_row_indices() take ONE index_select over the
    stacked tiles; anything else one index_copy_ per part.  (Separate from gather_tiles so that it is testable without
    a process group: the nccl branch of gather_tiles runs only on a multi-GPU node.)"""
    import torch
    world = tiling.world_size
    tile = scratch[0]
    if out is None:
        out = torch.empty((tiling.height, tiling.width) + tuple(tile.shape[2:]), dtype=tile.dtype, device=tile.device)
    base = getattr(scratch[0], "_rt_stack", None)
    perm = getattr(tiling, "_perm", None)
    if perm is not None and base is not None and base.shape[0] == world * tiling.max_rows and perm.device == base.device:
        # all tiles live in one allocation: de-interleave the frame with ONE gather kernel
        torch.index_select(base, 0, perm, out=out)
        return out
    for p in range(world):
        idx = tiling._idx[p] if (hasattr(tiling, "_idx") and tiling._idx[p].device == tile.device) else torch.as_tensor(tiling.rows[p], device=tile.device, dtype=torch.long)
        out.index_copy_(0, idx, scratch[p][: tiling.n_rows(p)])
    return out


def new_scratch(tile, world):
    """Receive buffers of the gather root: `world` tiles carved out of ONE allocation, so that the
    frame can be reassembled with a single index_select (see cache_row_indices)."""
    import torch
    base = torch.empty((world * tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    parts = list(base.split(tile.shape[0], dim=0))
    for p in parts:
        p._rt_stack = base
    return parts


def cache_row_indices(tiling: RowTiling, device):
    """Put the row index tensors on `device` once (avoids a host->device copy per frame):
    `_idx[p]` = image rows of part p; `_perm[y]` = position of image row y in the stacked tiles."""
    import torch
    tiling._idx = [torch.as_tensor(r, device=device, dtype=torch.long) for r in tiling.rows]
    perm = np.empty(tiling.height, dtype=np.int64)
    for p, rows in enumerate(tiling.rows):
        perm[np.asarray(rows, dtype=np.int64)] = p * tiling.max_rows + np.arange(len(rows), dtype=np.int64)
    tiling._perm = torch.as_tensor(perm, device=device)
    return tiling


def render_distributed(render_rows, tiling: RowTiling, rank, *, new_tile, dst=0, group=None, out=None, scratch=None):
    """One frame: `render_rows(part, tile)` fills this rank's tile (a [max_rows, W, 3] tensor made
    by `new_tile()` or reused), then the tiles are gathered to `dst`."""
    tile = new_tile()
    render_rows(rank, tile)
    return gather_tiles(tile, tiling, rank, dst=dst, group=group, out=out, scratch=scratch)
