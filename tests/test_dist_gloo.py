"""CPU test of the multi-GPU path with world_size 2 on the gloo backend: row tiling, padded gather
to rank 0 and de-interleave.  The per-rank renderer is the CPU oracle standing in for the HIP
renderer (same row-set contract: rt_frame.block_rows / part / n_parts)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_path


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, block_rows, result_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from raytracer_amd import load_scene
    from raytracer_amd.dist import RowTiling, render_distributed, cache_row_indices

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 2, 3)
        tiling = cache_row_indices(RowTiling(H, W, world, block_rows), "cpu")

        def render_rows(part, tile):
            img = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 11),
                                block_rows=block_rows, part=part, n_parts=world)
            tile[: img.shape[0]] = torch.from_numpy(img)

        frame = render_distributed(render_rows, tiling, rank,
                                   new_tile=lambda: torch.zeros((tiling.max_rows, W, 3), dtype=torch.float64))
        if rank == 0:
            np.save(result_path, frame.numpy())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,block_rows", [(2, 40, 30, 4), (2, 33, 21, 8), (3, 24, 22, 4)])
def test_tiled_render_on_ranks_equals_full_frame(tmp_path, oracle, world, W, H, block_rows):
    import torch.multiprocessing as mp
    from raytracer_amd import load_scene
    result = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, block_rows, result), nprocs=world, join=True)
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 2, 3)
    full = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 11))
    assert np.array_equal(np.load(result), full)


def test_row_tiling_partitions_all_rows():
    from raytracer_amd.dist import RowTiling
    for H, world, br in [(2160, 8, 16), (1080, 4, 16), (54, 3, 4), (7, 8, 2)]:
        t = RowTiling(H, 10, world, br)
        allrows = np.concatenate(t.rows)
        assert sorted(allrows) == list(range(H))
        assert t.max_rows == max(len(r) for r in t.rows)


def test_stacked_tiles_permutation_matches_per_part_copy():
    """The one-kernel de-interleave (index_select over the stacked receive buffers) and the
    per-part index_copy fallback build the same frame."""
    import torch
    from raytracer_amd.dist import RowTiling, cache_row_indices, new_scratch
    t = cache_row_indices(RowTiling(37, 5, 4, 3), "cpu")
    tile = torch.zeros((t.max_rows, 5, 3), dtype=torch.float64)
    parts = new_scratch(tile, 4)
    for p in range(4):
        parts[p].fill_(-1.0)
        for k, y in enumerate(t.rows[p]):
            parts[p][k] = float(y)
    a = torch.index_select(parts[0]._rt_stack, 0, t._perm)
    b = torch.empty_like(a)
    for p in range(4):
        b.index_copy_(0, t._idx[p], parts[p][: t.n_rows(p)])
    assert torch.equal(a, b) and torch.equal(a[:, 0, 0], torch.arange(37, dtype=torch.float64))


@pytest.mark.parametrize("world,H,block_rows", [(2, 50, 8), (3, 22, 4), (8, 2160, 8), (8, 61, 8)])
@pytest.mark.parametrize("dtype", ["float32", "float64", "uint8"])
def test_reassemble_stacked_and_separate_tiles(world, H, block_rows, dtype):
    """The gather root's de-interleave (one index_select over the stacked receive buffers, or one index_copy_ per part)
    for the tile formats bench.py gathers: float32 / float64 radiance and the 8-bit image (--out-format u8)."""
    import torch
    from raytracer_amd.dist import RowTiling, cache_row_indices, new_scratch, reassemble
    W = 5
    dt = getattr(torch, dtype)
    frame = (torch.arange(H * W * 3).reshape(H, W, 3) % 251).to(dt)
    for cached in (True, False):
        tiling = RowTiling(H, W, world, block_rows)
        if cached:
            cache_row_indices(tiling, "cpu")
        tile0 = torch.zeros((tiling.max_rows, W, 3), dtype=dt)
        for stacked in (True, False):
            scratch = new_scratch(tile0, world) if stacked else [torch.zeros_like(tile0) for _ in range(world)]
            for p in range(world):
                scratch[p][: tiling.n_rows(p)] = frame[torch.as_tensor(tiling.rows[p], dtype=torch.long)]
                scratch[p][tiling.n_rows(p):] = 77                       # padding rows must not reach the frame
            out = reassemble(scratch, tiling, torch.empty((H, W, 3), dtype=dt))
            assert torch.equal(out, frame), (cached, stacked)
            assert torch.equal(reassemble(scratch, tiling), frame)
