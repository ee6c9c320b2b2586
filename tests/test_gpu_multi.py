"""rt_render_multi (the multi-GPU boundary for callers without torch.distributed): one frame rendered by
several scene handles, interleaved row blocks, each handle copying its blocks straight into the frame.  One GPU is
enough to check the tiling, the strided copies (incl. a short last block) and the summed counters: the handles then
share the device.  On a multi-GPU box the same call spreads over the devices the handles were created on."""
import ctypes

import numpy as np
import pytest

from conftest import ROOT, rel_err, scene_path

pytestmark = pytest.mark.gpu


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


@pytest.mark.parametrize("n_handles,W,H,block_rows", [(3, 96, 70, 8), (2, 80, 48, 8), (4, 64, 37, 4), (5, 33, 9, 8)])
def test_multi_equals_single_fp64(oracle, n_handles, W, H, block_rows):
    from raytracer_amd import load_scene
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.render import GpuScene, render_multi
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 2, 3)
    cam.setup(W / H)
    packed = pack_scene(mats, surfs, lights, st.background_color)
    scenes = [GpuScene(packed) for _ in range(n_handles)]
    try:
        single = scenes[0].render(cam, W, H, 2, 3, precision="fp64", jitter=11)
        s1 = dict(scenes[0].last_stats)
        img, sm = render_multi(scenes, cam, W, H, 2, 3, precision="fp64", jitter=11, block_rows=block_rows)
        assert img.shape == (H, W, 3) and img.dtype == np.float64
        assert rel_err(img, single).max() < 1e-12           # (accumulation order of a pixel's child rays only)
        for k in ("rays", "hits", "shadow_rays", "segments", "level_rays", "level_hits"):
            assert sm[k] == s1[k], k
        ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 11))
        assert rel_err(img, ref).max() < 1e-9
    finally:
        for gs in scenes:
            gs.close()


def test_multi_device_output_fp32_and_u8(oracle):
    from raytracer_amd import load_scene
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.render import GpuScene, render_multi
    W, H = 160, 100
    cam, st, mats, surfs, lights = load_scene(scene_path("pool.txt"), 3, 2)
    cam.setup(W / H)
    packed = pack_scene(mats, surfs, lights, st.background_color)
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 5))
    hip = ctypes.CDLL("libamdhip64.so")
    scenes = [GpuScene(packed) for _ in range(3)]
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), W * H * 3 * 4) == 0
    try:
        _, sm = render_multi(scenes, cam, W, H, 3, 2, precision="fp32", jitter=5, out_ptr=buf.value)
        out = np.empty((H, W, 3), dtype=np.float32)
        assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), buf, out.nbytes, 2) == 0
        assert np.abs(_q(out) - _q(ref)).max() <= 1
        u8, _ = render_multi(scenes, cam, W, H, 3, 2, precision="fp32", jitter=5, out_format="u8")
        assert u8.dtype == np.uint8 and np.abs(u8.astype(int) - _q(ref)).max() <= 1
        single = scenes[0].render(cam, W, H, 3, 2, precision="fp32", jitter=5)
        assert sm["rays"] == scenes[0].last_stats["rays"] and sm["shadow_rays"] == scenes[0].last_stats["shadow_rays"]
        assert np.abs(out - single).max() < 2e-5
    finally:
        hip.hipFree(buf)
        for gs in scenes:
            gs.close()


def test_a_failing_handle_leaves_the_others_usable(monkeypatch):
    """handle 1 fails while handle 0 already holds its async frame pending with a pointer into the call's
    local stats array; the call must not leave that pointer behind (the next drain on handle 0 would write through it)."""
    from raytracer_amd import _lib, load_scene
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.render import GpuScene, render_multi
    W, H = 96, 64
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 2, 3)
    cam.setup(W / H)
    packed = pack_scene(mats, surfs, lights, st.background_color)
    scenes = [GpuScene(packed) for _ in range(3)]
    try:
        good, s_good = render_multi(scenes, cam, W, H, 2, 3, precision="fp64", jitter=3)
        good = good.copy()
        monkeypatch.setenv("RT_TEST_FAIL_HANDLE", "2")
        for _ in range(3):
            with pytest.raises(_lib.RtError) as ei:
                render_multi(scenes, cam, W, H, 2, 3, precision="fp64", jitter=3)
            assert "injected" in str(ei.value)
            junk = [np.full(4096, 7.0) for _ in range(8)]        # churn the host heap where the dead stats array lived
            for gs in scenes:
                single = gs.render(cam, W, H, 2, 3, precision="fp64", jitter=3)   # drains whatever the failed call left pending
                assert rel_err(single, good).max() < 1e-12
                assert gs.last_stats["rays"] == s_good["rays"]
            assert all(float(j[0]) == 7.0 and float(j.sum()) == 7.0 * 4096 for j in junk)
        monkeypatch.delenv("RT_TEST_FAIL_HANDLE")
        again, s_again = render_multi(scenes, cam, W, H, 2, 3, precision="fp64", jitter=3)
        assert rel_err(again, good).max() < 1e-12 and s_again["rays"] == s_good["rays"]
    finally:
        for gs in scenes:
            gs.close()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_two_frames_in_flight_give_the_single_frames(oracle, precision):
    """bench.py's throughput mode: frames alternate between two scene handles (own streams, own workspaces) with
    RT_FLAG_ASYNC, so that the kernels of two frames share the GPU.  Every frame must be the frame a blocking call on one
    handle renders (different jitter seeds per frame, so a mixed-up workspace would show), and match the oracle."""
    from raytracer_amd import _lib, load_scene
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.render import GpuScene
    W, H, n, depth = 384, 216, 3, 3
    cam, st, mats, surfs, lights = load_scene(scene_path("pool.txt" if precision == "fp32" else "original.txt"), n, depth)
    cam.setup(W / H)
    packed = pack_scene(mats, surfs, lights, st.background_color)
    item = 4 if precision == "fp32" else 8
    dt = np.float32 if precision == "fp32" else np.float64
    hip = ctypes.CDLL("libamdhip64.so")
    n_frames = 6
    bufs = []
    for _ in range(n_frames):
        b = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(b), W * H * 3 * item) == 0
        bufs.append(b)
    a, b2 = GpuScene(packed), GpuScene(packed)
    try:
        stats = [_lib.Stats() for _ in range(n_frames)]
        for k in range(n_frames):
            (a if k % 2 == 0 else b2).render(cam, W, H, n, depth, precision=precision, jitter=100 + k, out_ptr=bufs[k].value,
                                             flags=_lib.RT_FLAG_ASYNC, stats=stats[k])
        a.sync(); b2.sync()
        for k in range(n_frames):
            got = np.empty((H, W, 3), dtype=dt)
            assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), bufs[k], got.nbytes, 2) == 0
            single = a.render(cam, W, H, n, depth, precision=precision, jitter=100 + k)
            s1 = dict(a.last_stats)
            assert int(stats[k].rays) == s1["rays"] and int(stats[k].hits) == s1["hits"] and int(stats[k].shadow_rays) == s1["shadow_rays"]
            if precision == "fp64":
                assert rel_err(got, single).max() < 1e-12
            else:
                assert np.abs(got.astype(np.float64) - single.astype(np.float64)).max() < 1e-5      # (float atomics' order)
            if k < 2:
                ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 100 + k))
                if precision == "fp64":
                    assert rel_err(got, ref).max() < 1e-9
                else:
                    assert np.abs(_q(got) - _q(ref)).max() <= 1
    finally:
        a.close(); b2.close()
        for b in bufs:
            hip.hipFree(b)


def test_rccl_collectives_of_the_bench_on_a_raw_stream():
    """bench.py's N > 1 path issues torch.distributed.gather / all_reduce / all_gather / barrier on the nccl (= RCCL) backend
    with a raw hipStreamNonBlocking stream (torch.cuda.ExternalStream) current, then de-interleaves the received tiles on
    that stream.  No multi-GPU node is available to the tests, so this runs the same calls in a ONE-rank RCCL group on
    the one GPU of the box (in a child process: a process group is process-wide state): it cannot show scaling, but it
    does show that the backend, the external stream and the calls fit together on this stack."""
    import socket
    import subprocess
    import sys
    import textwrap
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        import torch, torch.distributed as dist
        from bench import new_stream
        from raytracer_amd.dist import RowTiling, cache_row_indices, gather_tiles, new_scratch
        torch.cuda.set_device(0); dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", device_id=dev)
        render, comm = new_stream(dev), new_stream(dev)
        tiling = cache_row_indices(RowTiling(40, 16, 1, 8), dev)
        for dt in (torch.float32, torch.float64, torch.uint8):
            with torch.cuda.stream(render):
                tile = (torch.arange(tiling.max_rows * 16 * 3, device=dev) % 251).reshape(tiling.max_rows, 16, 3).to(dt)
            ev = torch.cuda.Event(); ev.record(render)
            scratch = new_scratch(tile, 1)
            frame = torch.empty((40, 16, 3), dtype=dt, device=dev)
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                dist.gather(tile, gather_list=scratch, dst=0)          # (gather_tiles short-cuts a world of one)
                from raytracer_amd.dist import reassemble
                reassemble(scratch, tiling, frame)
            torch.cuda.synchronize()
            assert torch.equal(frame, tile), dt
        t = torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM)
        parts = [torch.zeros_like(t)]
        dist.all_gather(parts, t)
        assert torch.equal(parts[0], t)
        dist.barrier(); torch.cuda.synchronize()
        dist.destroy_process_group()
        print("rccl ok")
    """)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200, cwd=ROOT)
    assert res.returncode == 0 and "rccl ok" in res.stdout, (res.stdout[-500:], res.stderr[-2000:])
