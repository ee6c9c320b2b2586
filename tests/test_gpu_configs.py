"""BASELINE.json configurations and SURVEY.md §8(f) rows that had no GPU-side pin of their own:

* configs[0] (pool.txt 500x500, file settings n=5 D=10): the facts the reference itself produced in the
  build container with `np.random.seed(1234)` (381,520 rays, 39,822,500 shadow rays, 11
  batches of known sizes; ray_tracer.py:477-489, 602) must come out of the GPU's numpy-stream mode;
* configs[3] (synthetic 4096 spheres + 512 boxes, the only config with boxes at scale, cube.py:64-136):
  the exact scene at 96x54 against the oracle, and at 3840x2160 through size-independent properties;
* f1 (save_image quantiser on the device, ray_tracer.py:606-608): `out_format="u8"` against the quantised
  images the REFERENCE rendered (golden image_B), not against the GPU's own float image;
* f3 (parser hardening + packing cache, ray_tracer.py:39-91): renders from `load_packed()` output;
* e (multi-GPU): bench.py's N = 2 code path (row tiling, gather, verification) rehearsed on one device.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err, scene_path

pytestmark = pytest.mark.gpu


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


def _synth(n_sph=4096, n_box=512, n=3, depth=3):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.synth import synthetic_scene_text
    cam, st, objs = parse_scene_text(synthetic_scene_text(n_sph, n_box, shadow_rays=n, max_recursions=depth))   # seed 20260213
    mats, surfs, lights = separate_objects(objs)
    return cam, st, mats, surfs, lights


# ------------------------------------------------------------------------------- configs[0] ----
def test_config1_pool_500_reference_run_facts(oracle):
    from raytracer_amd import load_scene
    from raytracer_amd.render import render_hip
    W = H = 500
    cam, st, mats, surfs, lights = load_scene(scene_path("pool.txt"))          # file settings: 5x5 shadow rays, depth 10
    assert (st.root_number_shadow_rays, st.max_recursions) == (5, 10)
    stats = {}
    img = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp64", jitter=("numpy", 1234), stats=stats, verbose=False)
    assert stats["rays"] == 381520                                             # "total rays" the reference printed
    assert stats["shadow_rays"] == 39822500
    assert stats["batches"] == 11
    assert stats["level_rays"] == [250000, 95652, 20690, 7722, 3494, 1712, 934, 564, 356, 232, 164]
    o_img, o_stats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("numpy", 1234), return_stats=True)
    assert rel_err(img, o_img).max() < 1e-9
    for k in ("rays", "hits", "shadow_rays", "segments", "batches"):
        assert stats[k] == o_stats[k], k
    # depth 10 through the fast mode as well (pixel-keyed jitter): same ray tree, every channel within one level
    s32 = {}
    a = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp32", jitter=("pixel", 1234), stats=s32, verbose=False)
    b = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 1234))
    assert s32["rays"] == 381520 and s32["levels"] == 11
    assert np.abs(_q(a) - _q(b)).max() <= 1


# ------------------------------------------------------------------------------- configs[3] ----
def test_config4_scene_96x54_vs_oracle(oracle):
    from raytracer_amd.render import render_hip
    cam, st, mats, surfs, lights = _synth()
    assert len(surfs) == 4609
    W, H = 96, 54
    s64, s32 = {}, {}
    img = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp64", jitter=("pixel", 1234), stats=s64, verbose=False)
    ref, o_stats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 1234), return_stats=True)
    assert rel_err(img, ref).max() < 1e-9
    for k in ("rays", "hits", "shadow_rays", "segments"):
        assert s64[k] == o_stats[k], k
    img32 = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp32", jitter=("pixel", 1234), stats=s32, verbose=False)
    for k in ("rays", "hits", "shadow_rays"):
        assert s32[k] == o_stats[k], k
    d8 = np.abs(_q(img32) - _q(ref))
    print(f"config 4 scene 96x54 fp32: max 8-bit diff {d8.max()}, {s32['retested']} (hit, light) pairs re-tested in fp64")
    assert d8.max() <= 1


def test_config4_full_size_properties():
    """3840x2160: fp32 against the library's fp64 mode (the one pinned to the reference), identical ray trees,
    and the 8 row tiles of the multi-GPU partition reassemble the frame."""
    from raytracer_amd.packing import pack_scene, rows_in_part
    from raytracer_amd.render import GpuScene
    cam, st, mats, surfs, lights = _synth()
    W, H = 3840, 2160
    cam.setup(W / H)
    with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
        a = gs.render(cam, W, H, 3, 3, precision="fp32", jitter=1234); sa = dict(gs.last_stats)
        b = gs.render(cam, W, H, 3, 3, precision="fp64", jitter=1234); sb = dict(gs.last_stats)
        for k in ("rays", "hits", "shadow_rays", "level_rays", "level_hits"):
            assert sa[k] == sb[k], k
        assert sa["rays"] > 1.5 * W * H and sa["levels"] == 4
        d8 = np.abs(_q(a) - _q(b))
        print(f"config 4 4K fp32 vs fp64: max 8-bit diff {d8.max()}, channels > 1 level {(d8 > 1).sum()} of {d8.size}, "
              f"{sa['retested']} pairs re-tested")
        assert d8.max() <= 1
        out = np.empty_like(a)
        for part in range(8):
            out[rows_in_part(H, 8, part, 8)] = gs.render(cam, W, H, 3, 3, precision="fp32", jitter=1234, block_rows=8, part=part, n_parts=8)
        assert np.abs(out - a).max() < 2e-5                 # accumulation order of a pixel's child rays only
        assert np.abs(_q(out) - _q(b)).max() <= 1


# ----------------------------------------------------------------------------------- f1, f3 ----
@pytest.mark.parametrize("name", ["g1_pool_64", "g2_original_96x54", "g3_boxes_80x60", "g7_synth_small_48x27",
                                  "g8_pool_n1_40", "g9_original_d0_48x27"])
@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_u8_output_against_the_references_quantised_image(name, precision):
    from raytracer_amd import load_scene
    from raytracer_amd.render import GpuScene
    g = load_golden(name)
    W, H = int(g["width"]), int(g["height"])
    cam, st, mats, surfs, lights = load_scene(scene_path(str(g["scene"])), int(g["n_shadow"]), int(g["max_depth"]))
    cam.setup(W / H)
    with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
        u8 = gs.render(cam, W, H, int(g["n_shadow"]), int(g["max_depth"]), precision=precision, jitter=int(g["seed"]), out_format="u8")
    assert u8.dtype == np.uint8 and u8.shape == (H, W, 3)
    want = _q(g["image_B"])                                # clip, x255, truncate of the reference's own image
    d = np.abs(u8.astype(int) - want)
    print(f"{name} {precision} u8: {(d != 0).sum()} of {d.size} channels differ (values at a bin edge), max {d.max()}")
    assert d.max() <= 1
    if precision == "fp64":
        assert (d != 0).mean() < 1e-3                      # 1e-16 differences only matter exactly at bin edges


@pytest.mark.parametrize("name,n,depth", [("pool.txt", 3, 2), ("boxes.txt", None, None), ("synth_small.txt", None, None)])
def test_render_from_load_packed(tmp_path, oracle, name, n, depth):
    from raytracer_amd import load_scene
    from raytracer_amd.render import GpuScene
    from raytracer_amd.scene_cache import load_packed
    W, H = 72, 48
    cam_o, st_o, mats, surfs, lights = load_scene(scene_path(name), n, depth)                # object route, for the oracle
    n_sh, dep = int(st_o.root_number_shadow_rays), int(st_o.max_recursions)
    ref = oracle.render(cam_o, st_o, mats, surfs, lights, W, H, jitter=("pixel", 9))
    for _ in range(2):                                                                       # second pass: from the disk cache
        cam, st, packed = load_packed(scene_path(name), cache_dir=str(tmp_path))
        cam.setup(W / H)
        with GpuScene(packed) as gs:
            img = gs.render(cam, W, H, n_sh, dep, precision="fp64", jitter=9)
            img32 = gs.render(cam, W, H, n_sh, dep, precision="fp32", jitter=9)
        assert rel_err(img, ref).max() < 1e-9
        assert np.abs(_q(img32) - _q(ref)).max() <= 1


# --------------------------------------------------------------------------- e: N = 2 rehearsal ----
@pytest.mark.parametrize("pipeline,fmt", [("1", "native"), ("0", "native"), ("1", "u8")])
def test_bench_two_ranks_on_one_device(pipeline, fmt):
    """bench.py's N > 1 path (interleaved row blocks per rank, gather to rank 0, reassembly) with both ranks on
    cuda:0 and the gather through gloo: the gathered frame must equal the single-GPU frame exactly (fp64 workload:
    every pixel receives at most one contribution per level, so there is no accumulation-order rounding)."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_ONE_DEVICE="1", RT_BENCH_PIPELINE=pipeline, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--prewarm", "2", "--workload", "pool500", "--out-format", fmt]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["verify"]["ranks"] == 2
    assert out["verify"]["max_abs_diff_vs_single_gpu_frame"] == 0.0
    assert out["config"]["radiance_rays"] == 381520        # both ranks' rays add up to the frame's
    # the keys a first hardware run is read from: per-rank compute vs gather time
    m = out["multi"]
    assert len(m["compute_ms"]["per_rank"]) == 2 and len(m["gather_ms"]["per_rank"]) == 2 and m["frames_with_events"] >= 1
    assert m["compute_ms"]["max"] >= m["compute_ms"]["min"] > 0.0 and m["gather_ms"]["rank0"] > 0.0
    assert sum(m["rows_per_rank"]) == 500 and m["pipelined"] == (pipeline == "1")
    assert m["frames_in_flight"] == (3 if pipeline == "1" else 1)
    assert out["config"]["out_format"] == ("u8" if fmt == "u8" else "f64")
