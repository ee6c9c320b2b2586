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
