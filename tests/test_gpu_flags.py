"""Frame flags and queue sizing that no other test pins:
* RT_FLAG_BRUTE_FORCE (bench.py's roofline basis): every culling step and shortcut off — images and counters must be
  those of the culled frame (fp64: to accumulation-order rounding; fp32: the same, both within one level of the oracle);
* the starting size of the bounce queues (RT_QUEUE_SLOTS_PER_PIXEL) and what happens when it is too small: a blocking
  call grows the queues and renders again by itself, an RT_FLAG_ASYNC frame reports RT_ERR_QUEUE_OVERFLOW at rt_sync and
  the next frame gets the larger queues."""
import ctypes
import os

import numpy as np
import pytest

from conftest import rel_err, scene_path

pytestmark = pytest.mark.gpu


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


@pytest.mark.parametrize("scene,n,depth", [("pool.txt", 3, 2), ("boxes.txt", None, None), ("synth_small.txt", None, None)])
def test_brute_force_flag_renders_the_same_frame(oracle, scene, n, depth):
    from raytracer_amd import _lib, load_scene
    from raytracer_amd.render import GpuScene
    W, H = 96, 64
    cam, st, mats, surfs, lights = load_scene(scene_path(scene), n, depth)
    n_sh, dep = int(st.root_number_shadow_rays), int(st.max_recursions)
    cam.setup(W / H)
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 21))
    with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
        for prec in ("fp64", "fp32"):
            a = gs.render(cam, W, H, n_sh, dep, precision=prec, jitter=21); sa = dict(gs.last_stats)
            b = gs.render(cam, W, H, n_sh, dep, precision=prec, jitter=21, flags=_lib.RT_FLAG_BRUTE_FORCE); sb = dict(gs.last_stats)
            for k in ("rays", "hits", "shadow_rays", "level_rays", "level_hits"):
                assert sa[k] == sb[k], (prec, k)
            if prec == "fp64":
                assert rel_err(a, b).max() < 1e-12 and rel_err(b, ref).max() < 1e-9
            else:
                assert np.abs(_q(b) - _q(ref)).max() <= 1 and np.abs(_q(a) - _q(ref)).max() <= 1
                assert np.abs(a - b).max() < 1e-4        # the same samples decided the same way; fp32 sums in another order


OVERFLOW_SCENE = ("cam 0 0 0 0 0 -1 0 1 0 1 2\nset 0.2 0.3 0.4 1 4\n"
                  "mtl 0.5 0.5 0.5 0.5 0.5 0.5 0.4 0.4 0.4 10 0.5\n"
                  "sph 0 0 0 2 1\nsph 0 0 0 3 1\nsph 0 0 0 4 1\n"
                  "lgt 0 0 0 1 1 1 1 0.5 1\n")


def test_small_queues_blocking_retry_and_async_report(oracle, monkeypatch):
    from raytracer_amd import _lib, parse_scene_text, separate_objects
    from raytracer_amd.render import GpuScene
    cam, st, objs = parse_scene_text(OVERFLOW_SCENE)
    mats, surfs, lights = separate_objects(objs)
    W, H = 320, 200                      # 64,000 pixels: the constant part of the queue size (65,536 slots) does not cover 4 levels of doubling
    cam.setup(W / H)
    ref, ostats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 4), return_stats=True)
    monkeypatch.setenv("RT_QUEUE_SLOTS_PER_PIXEL", "0.05")
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), W * H * 3 * 8) == 0
    try:
        # an async frame cannot be replayed: the overflow is reported at rt_sync ...
        with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
            gs.render(cam, W, H, 1, 4, precision="fp64", jitter=4, out_ptr=buf.value, flags=_lib.RT_FLAG_ASYNC)
            with pytest.raises(_lib.RtError) as e:
                gs.sync()
            assert e.value.code == _lib.RT_ERR_QUEUE_OVERFLOW
            # ... and the queues have been grown for the frames that follow
            for _ in range(4):
                gs.render(cam, W, H, 1, 4, precision="fp64", jitter=4, out_ptr=buf.value, flags=_lib.RT_FLAG_ASYNC)
                try:
                    gs.sync()
                    break
                except _lib.RtError as err:                     # levels below the truncated one were undercounted: grows again
                    assert err.code == _lib.RT_ERR_QUEUE_OVERFLOW
            else:
                pytest.fail("queues did not reach the needed size in 5 async frames")
            out = np.empty((H, W, 3))
            assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), buf, out.nbytes, 2) == 0
            assert rel_err(out, ref).max() < 1e-9
        # a blocking frame grows and retries by itself
        with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
            img = gs.render(cam, W, H, 1, 4, precision="fp64", jitter=4)
            assert rel_err(img, ref).max() < 1e-9
            assert gs.last_stats["rays"] == ostats["rays"] and max(gs.last_stats["level_rays"]) > 1.5 * W * H
    finally:
        hip.hipFree(buf)


def test_level0_fusion_switches_change_nothing(monkeypatch):
    """Level 0 fused with the nearest-hit query (k_shade_ls: default; k_shade_exact_items: RT_FUSE0_FP64=1) against the
    two-kernel level 0 (RT_FUSE0=0): same rays, same hits, same pixels — a frame with background tiles, partial tiles
    and a ragged right / bottom edge."""
    import numpy as np
    from conftest import scene_path
    from raytracer_amd import load_scene
    from raytracer_amd.render import GpuScene
    from raytracer_amd import parse_scene_text, separate_objects
    text = """cam 0 1 7  0 0.5 0  0 1 0  1.2 2.4
set 0.1 0.2 0.4 3 3
mtl 0.7 0.3 0.2 0.6 0.6 0.6 0.3 0.3 0.3 20 0.0
mtl 0.2 0.6 0.8 0.9 0.9 0.9 0.1 0.1 0.1 50 0.5
mtl 0.6 0.6 0.6 0.3 0.3 0.3 0.2 0.2 0.2 5 0.0
sph -1.2 0.5 0 1.0 1
sph 1.0 0.8 0.5 0.8 2
sph 0.2 -0.4 1.5 0.4 3
box 2.5 0.0 -1.0 1.0 3
lgt 3 5 4 0.9 0.9 0.8 1 0.8 1.0
lgt -4 3 2 0.4 0.5 0.7 1 0.6 0.5
"""
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    W, H = 203, 77
    cam.setup(W / H)
    with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
        out = {}
        for name, env in (("fused", {}), ("split", {"RT_FUSE0": "0"}), ("fused64", {"RT_FUSE0_FP64": "1"})):
            for k in ("RT_FUSE0", "RT_FUSE0_FP64"):
                monkeypatch.delenv(k, raising=False)
            for k, v_ in env.items():
                monkeypatch.setenv(k, v_)
            a32 = gs.render(cam, W, H, 3, 3, precision="fp32", jitter=21).copy(); s32 = dict(gs.last_stats)
            a64 = gs.render(cam, W, H, 3, 3, precision="fp64", jitter=21).copy(); s64 = dict(gs.last_stats)
            out[name] = (a32, s32, a64, s64)
        for name in ("split", "fused64"):
            for k in ("rays", "hits", "shadow_rays", "level_rays", "level_hits"):
                assert out[name][1][k] == out["fused"][1][k], (name, k)
                assert out[name][3][k] == out["fused"][3][k], (name, k)
            assert out[name][3]["segments"] == out["fused"][3]["segments"]
            assert np.abs(out[name][0] - out["fused"][0]).max() < 2e-6          # (order of a pixel's atomic adds at levels >= 1)
            assert np.abs(out[name][2] - out["fused"][2]).max() < 1e-13
        assert out["fused"][1]["level_hits"][0] < W * H                        # the frame has background pixels
