"""GPU parity for RT_FLAG_SEQUENTIAL: the wavefront kernels run with the semantics
of the reference's per-pixel recursive `render` (ray_tracer.py:296-434) and are compared with
(a) images produced by that reference function itself (tests/golden/gs*.npz, pixel-keyed jitter) and
(b) the CPU oracle's literal recursive restatement."""
import numpy as np
import pytest

from conftest import load_golden, rel_err, scene_path

pytestmark = pytest.mark.gpu

SEQ_CASES = ["gs1_original_48x27", "gs2_pool_32", "gs3_boxes_40x30", "gs4_original_d1_40x24", "gs5_original_n1_d0_32x18"]


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


@pytest.mark.parametrize("name", SEQ_CASES)
def test_sequential_fp64_matches_reference_render(name):
    from raytracer_amd import load_scene
    from raytracer_amd.render import render
    g = load_golden(name)
    cam, st, mats, surfs, lights = load_scene(scene_path(str(g["scene"])), int(g["n_shadow"]), int(g["max_depth"]))
    img = render(cam, st, mats, surfs, lights, int(g["width"]), int(g["height"]), jitter=("pixel", int(g["seed"])), verbose=False)
    err = rel_err(img, g["image_B"]).max()
    print(f"{name}: GPU sequential fp64 vs reference render(): max rel err {err:.2e}")
    assert err < 1e-9


@pytest.mark.parametrize("name", SEQ_CASES)
def test_sequential_fp32_matches_reference_render(name):
    from raytracer_amd import load_scene
    from raytracer_amd.render import render
    g = load_golden(name)
    cam, st, mats, surfs, lights = load_scene(scene_path(str(g["scene"])), int(g["n_shadow"]), int(g["max_depth"]))
    img = render(cam, st, mats, surfs, lights, int(g["width"]), int(g["height"]), jitter=("pixel", int(g["seed"])),
                 precision="fp32", verbose=False)
    d8 = np.abs(_q(img) - _q(g["image_B"]))
    print(f"{name}: GPU sequential fp32: max 8-bit diff {d8.max()}, channels > 1 level {(d8 > 1).sum()} of {d8.size}")
    assert d8.max() <= 1, f"{(d8 > 1).sum()} channels off by more than one level"


def test_sequential_fp64_vs_oracle_larger_frame(oracle):
    from raytracer_amd import load_scene
    from raytracer_amd.render import render
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 3, 4)
    stats = {}
    img = render(cam, st, mats, surfs, lights, 320, 180, jitter=("pixel", 9), verbose=False, stats=stats)
    ref, ost = oracle.render(cam, st, mats, surfs, lights, 320, 180, jitter=("pixel", 9), sequential=True, return_stats=True)
    assert stats["rays"] == ost["rays"] and stats["hits"] == ost["hits"] and stats["shadow_rays"] == ost["shadow_rays"]
    assert rel_err(img, ref).max() < 1e-9


def test_sequential_rejects_numpy_stream():
    from raytracer_amd import load_scene, _lib
    from raytracer_amd.render import render
    cam, st, mats, surfs, lights = load_scene(scene_path("pool.txt"), 2, 1)
    with pytest.raises(_lib.RtError):
        render(cam, st, mats, surfs, lights, 16, 16, jitter=("numpy", 1), verbose=False)


@pytest.mark.parametrize("seed", [0, 1, 4, 15, 18, 49, 52])
def test_sequential_fp32_random_scenes(oracle, seed):
    """Scenes of tools/sequential_survey.py that the fp32 kernels got wrong before they knew the sequential renderer's
    specular rule (term only for r.v > 0, ray_tracer.py:356-359: with shininess 0 the vectorized renderer adds
    Ks*Lc*1 where r.v = 0, the sequential one nothing — up to 240 levels on whole surfaces) and before the re-test
    walk took the sequential walk's rules (no 20-step cap)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from test_gpu_random_scenes import random_scene_text
    from test_gpu_many_spheres import many_spheres_text
    from authored_survey import authored_scene_text
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    kind = seed % 3
    text = random_scene_text(7000 + seed, seed % 2 == 0) if kind == 0 else (authored_scene_text(seed) if kind == 1 else many_spheres_text(7 + seed % 10, 9000 + seed, seed % 2 == 1))
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    ref = oracle.render(cam, st, mats, surfs, lights, 64, 48, jitter=("pixel", seed), sequential=True)
    i64 = render_hip(cam, st, mats, surfs, lights, 64, 48, jitter=("pixel", seed), sequential=True, verbose=False)
    i32 = render_hip(cam, st, mats, surfs, lights, 64, 48, jitter=("pixel", seed), sequential=True, precision="fp32", verbose=False)
    assert (np.abs(i64 - ref) / np.maximum(np.abs(ref), 1e-3)).max() < 1e-9
    d8 = np.abs(_q(i32) - _q(ref))
    print(f"seed {seed}: {len(surfs)} surfaces, sequential fp32 max 8-bit diff {d8.max()}")
    assert d8.max() <= 1
