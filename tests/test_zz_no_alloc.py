"""rt_render_static — the entry point that allocates nothing (include/rt_hip.h) — against rt_render,
through the C-ABI, on two calls in a row (the second with a smaller input, so stale contents of the
shared static scratch array would show), plus its size limit.  Collected last on purpose."""
import ctypes

import numpy as np
import pytest

from conftest import rel_err, scene_path

pytestmark = pytest.mark.gpu


class DeviceBuffer:
    def __init__(self, hip, nbytes):
        self.hip, self.ptr, self.nbytes = hip, ctypes.c_void_p(), nbytes
        assert hip.hipMalloc(ctypes.byref(self.ptr), nbytes) == 0

    def fill(self, byte):
        assert self.hip.hipMemset(self.ptr, byte, self.nbytes) == 0

    def to_host(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        assert self.hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes, 2) == 0   # device -> host
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def free_device_bytes(hip):
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_static_entry_matches_rt_render_on_two_calls_in_a_row():
    from raytracer_amd import load_scene
    from raytracer_amd.image import quantize_u8
    from raytracer_amd.render import GpuScene
    hip = ctypes.CDLL("libamdhip64.so")
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 3, 4)    # reflection + transparency children
    calls = [(160, 96, "fp64", "f64", np.float64, 8), (72, 40, "fp32", "f32", np.float32, 4), (72, 40, "fp64", "u8", np.uint8, 1)]
    a = DeviceBuffer(hip, 160 * 96 * 3 * 8)
    b = DeviceBuffer(hip, 160 * 96 * 3 * 8)
    try:
        with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
            refs = []
            for W, H, prec, fmt, dtype, _ in calls:          # rt_render first: its workspace is grown before memory is measured
                cam.setup(W / H)
                a.fill(0xEE)
                gs.render(cam, W, H, 3, 4, precision=prec, jitter=11, out_format=fmt, out_ptr=a.ptr.value)
                refs.append((a.to_host((H, W, 3), dtype), dict(gs.last_stats)))
            before = free_device_bytes(hip)
            for (W, H, prec, fmt, dtype, _), (ref, ref_stats) in zip(calls, refs):
                cam.setup(W / H)
                b.fill(0xEE)
                gs.render(cam, W, H, 3, 4, precision=prec, jitter=11, out_format=fmt, out_ptr=b.ptr.value, static=True)
                got = b.to_host((H, W, 3), dtype)
                stats = gs.last_stats
                assert stats["rays"] > W * H and stats["hits"] > 0 and stats["levels"] == 5
                for k in ("rays", "hits", "shadow_rays", "segments", "level_rays", "level_hits"):
                    assert stats[k] == ref_stats[k], k
                if fmt == "f64":       # the only difference allowed: the order of a pixel's atomic adds at levels >= 1
                    assert rel_err(got, ref).max() < 1e-12
                elif fmt == "f32":
                    assert np.abs(quantize_u8(got).astype(int) - quantize_u8(ref).astype(int)).max() <= 1
                    assert np.abs(got - ref).max() < 2e-6
                else:
                    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1
                    assert (got != ref).mean() < 1e-3
            assert free_device_bytes(hip) == before          # nothing was allocated (or freed) on the device
    finally:
        a.free(); b.free()


def test_static_entry_rejects_what_it_cannot_serve():
    from raytracer_amd import _lib, load_scene
    from raytracer_amd.render import GpuScene
    hip = ctypes.CDLL("libamdhip64.so")
    cam, st, mats, surfs, lights = load_scene(scene_path("pool.txt"), 2, 1)
    buf = DeviceBuffer(hip, 1024)
    try:
        with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
            cam.setup(1.0)
            with pytest.raises(_lib.RtError, match="at most"):
                gs.render(cam, 512, 512, 2, 1, precision="fp32", out_ptr=buf.ptr.value, static=True)   # > RT_STATIC_MAX_PIXELS
            with pytest.raises(_lib.RtError, match="pixel-keyed"):
                gs.render(cam, 8, 8, 2, 1, precision="fp64", jitter=("numpy", 1), out_ptr=buf.ptr.value, static=True)
            with pytest.raises(_lib.RtError, match="RT_FLAG_TIMING"):
                gs.render(cam, 8, 8, 2, 1, precision="fp64", out_ptr=buf.ptr.value, static=True, flags=_lib.RT_FLAG_TIMING)
            # a row set of a large frame is fine: 512x512 in 4 parts of 64-row blocks
            big = DeviceBuffer(hip, 512 * 128 * 3 * 4)
            try:
                gs.render(cam, 512, 512, 2, 1, precision="fp32", block_rows=64, part=1, n_parts=4, out_ptr=big.ptr.value, static=True)
                assert gs.last_stats["rays"] >= 512 * 128
            finally:
                big.free()
    finally:
        buf.free()


def test_static_entry_runs_under_graph_capture_and_reports_async_overflow():
    """hipMemGetInfo cannot see a hipMalloc paired with a hipFree, pinned allocations or event creation.  A stream
    capture in GLOBAL mode can: any allocation, synchronisation or event creation inside the captured region invalidates
    the capture and the call fails.  The async static frame is captured, replayed twice and compared with rt_render;
    then an async static frame that must overflow its queues is reported by rt_sync (it used to be lost)."""
    import torch
    from raytracer_amd import _lib, load_scene
    from raytracer_amd.render import GpuScene
    dev = torch.device("cuda", 0)
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 3, 4)
    W, H = 120, 72
    cam.setup(W / H)
    ref = torch.empty((H, W, 3), dtype=torch.float64, device=dev)
    out = torch.full((H, W, 3), -7.0, dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream(dev)
    with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
        gs.render(cam, W, H, 3, 4, precision="fp64", jitter=5, out_ptr=ref.data_ptr())
        ref_stats = dict(gs.last_stats)
        gs.render(cam, W, H, 3, 4, precision="fp64", jitter=5, out_ptr=out.data_ptr(), static=True)      # (first use outside the capture)
        assert gs.last_stats["rays"] == ref_stats["rays"]
        out.fill_(-7.0)
        torch.cuda.synchronize()
        gs.set_stream(stream.cuda_stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="global"):
            gs.render(cam, W, H, 3, 4, precision="fp64", jitter=5, out_ptr=out.data_ptr(), static=True, flags=_lib.RT_FLAG_ASYNC)
        torch.cuda.synchronize()
        assert float(out.min()) == -7.0                       # captured, not executed
        for _ in range(2):
            out.fill_(-7.0)
            g.replay()
            torch.cuda.synchronize()
            assert float(((out - ref).abs() / ref.abs().clamp_min(1e-12)).max()) < 1e-12
        del g
        gs.set_stream(None)
        gs.sync()                                             # the captured frame's counters: no overflow
    # An async static frame that cannot fit: the camera sits inside a sphere that is transparent AND reflective, inside a
    # larger opaque one, so every pixel's hit spawns two children that both hit something: level 1 holds 2 rays (and 2
    # hits) per pixel, the static queues ~1.3 at the maximum row-set size.
    from raytracer_amd import parse_scene_text, separate_objects
    text = """cam 0 0 0  0 0 1  0 1 0  1 2
set 0.2 0.3 0.4 1 3
mtl 0.5 0.5 0.5 1 1 1 0.5 0.5 0.5 10 0.5
mtl 0.4 0.6 0.5 0.5 0.5 0.5 0 0 0 10 0
sph 0 0 0 10 1
sph 0 0 0 20 2
lgt 0 5 0 1 1 1 1 0.5 1
"""
    cam2, st2, objs = parse_scene_text(text)
    mats2, surfs2, lights2 = separate_objects(objs)
    cam2.setup(512 / 256)
    big = torch.empty((256 * 512 * 3,), dtype=torch.float32, device=dev)
    with GpuScene.from_objects(mats2, surfs2, lights2, st2.background_color) as gs:
        with pytest.raises(_lib.RtError) as ei:
            gs.render(cam2, 512, 256, 1, 3, precision="fp32", jitter=5, out_ptr=big.data_ptr(), static=True)
        assert ei.value.code == _lib.RT_ERR_QUEUE_OVERFLOW
        gs.render(cam2, 512, 256, 1, 3, precision="fp32", jitter=5, out_ptr=big.data_ptr(), static=True, flags=_lib.RT_FLAG_ASYNC)
        with pytest.raises(_lib.RtError) as ei:
            gs.sync()                                         # used to return RT_OK with an incomplete image
        assert ei.value.code == _lib.RT_ERR_QUEUE_OVERFLOW
        gs.sync()                                             # reported once
        gs.render(cam2, 512, 256, 1, 3, precision="fp32", jitter=5, out_ptr=big.data_ptr())      # rt_render grows and retries
        assert gs.last_stats["level_rays"][1] == 2 * 512 * 256
