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


@pytest.mark.parametrize("W,H,block_rows", [(40, 30, 4), (33, 21, 8)])
def test_two_rank_tiled_render_equals_full_frame(tmp_path, oracle, W, H, block_rows):
    import torch.multiprocessing as mp
    from raytracer_amd import load_scene
    result = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), W, H, block_rows, result), nprocs=2, join=True)
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
