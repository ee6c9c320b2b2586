"""Large scenes (> 128 primitives) go through the culling hierarchy (rt_device.hpp
BvhNode / nearest_accel, per-lane walks in k_shade_fast).  The hierarchy must never change a result:
every check here compares with the brute-force CPU oracle and with the library's own brute-force
mode (RT_ACCEL=0)."""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _scene(n_sph, n_box, seed=3, n=3, depth=3):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.synth import synthetic_scene_text
    cam, st, objs = parse_scene_text(synthetic_scene_text(n_sph, n_box, shadow_rays=n, max_recursions=depth, seed=seed))
    mats, surfs, lights = separate_objects(objs)
    return cam, st, mats, surfs, lights


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


def test_nearest_hit_large_scene_bit_exact(oracle):
    from raytracer_amd.render import nearest_hit
    cam, st, mats, surfs, lights = _scene(1500, 300)
    rng = np.random.default_rng(11)
    M = 20000
    o = rng.uniform([-60, -2, -60], [60, 30, 60], (M, 3))
    tgt = rng.uniform([-48, -0.5, -48], [48, 12, 48], (M, 3))
    d = tgt - o
    d[: M // 2] /= np.linalg.norm(d[: M // 2], axis=1, keepdims=True)      # second half stays unnormalised
    # rays starting inside primitives and on their surfaces
    c = np.array([s.position for s in surfs[1:400]])
    o[:399] = c + rng.normal(scale=0.05, size=c.shape)
    t, idx, n = nearest_hit(surfs, o, d)
    to, io, no = oracle.nearest_hit(surfs, o, d)
    assert np.array_equal(idx, io) and np.array_equal(t, to) and np.array_equal(n, no)
    assert (idx >= 0).mean() > 0.5


def test_shadow_transmission_large_scene_bit_exact(oracle):
    from raytracer_amd.render import shadow_transmission
    cam, st, mats, surfs, lights = _scene(1200, 200)
    rng = np.random.default_rng(5)
    M = 8000
    o = rng.uniform([-48, -0.9, -48], [48, 12, 48], (M, 3))
    L = np.array([30.0, 40.0, 30.0]) + rng.normal(scale=1.0, size=(M, 3))
    d = L - o
    dist = np.linalg.norm(d, axis=1)
    d /= dist[:, None]
    a = shadow_transmission(mats, surfs, o, d, dist)
    b = oracle.shadow_transmission(mats, surfs, o, d, dist)
    assert np.array_equal(a, b)
    assert 0.05 < (a < 1).mean() < 0.95


@pytest.mark.parametrize("n_sph,n_box,W,H", [(600, 80, 96, 54), (2000, 0, 64, 36), (0, 300, 64, 36)])
def test_render_large_scene_fp64_and_fp32(oracle, n_sph, n_box, W, H):
    from raytracer_amd.render import render_hip
    cam, st, mats, surfs, lights = _scene(n_sph, n_box)
    stats = {}
    img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 21), stats=stats, verbose=False)
    ref, ost = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 21), return_stats=True)
    for k in ("rays", "hits", "shadow_rays", "segments"):
        assert stats[k] == ost[k], k
    assert rel_err(img, ref).max() < 1e-12
    img32 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 21), precision="fp32", verbose=False)
    d8 = np.abs(_q(img32) - _q(ref))
    print(f"{n_sph} spheres + {n_box} boxes {W}x{H} fp32: max 8-bit diff {d8.max()}, channels > 1 level {(d8 > 1).sum()} of {d8.size}")
    assert d8.max() <= 1, f"{(d8 > 1).sum()} channels off by more than one level"


def test_hierarchy_equals_brute_force_mode():
    """Same library, RT_ACCEL=0 (brute force over all records) vs the hierarchy.  Every hit and every
    shadow term is bit-identical; the images agree to the rounding of the framebuffer's atomic
    accumulation order (deeper levels add to a pixel in queue order, which is not deterministic)."""
    from raytracer_amd.render import render_hip
    cam, st, mats, surfs, lights = _scene(900, 150, seed=8, n=2, depth=2)
    out = {}
    for mode in ("1", "0"):
        os.environ["RT_ACCEL"] = mode
        try:
            out[mode] = (render_hip(cam, st, mats, surfs, lights, 80, 45, jitter=("pixel", 2), verbose=False),
                         render_hip(cam, st, mats, surfs, lights, 80, 45, jitter=("pixel", 2), precision="fp32", verbose=False))
        finally:
            os.environ.pop("RT_ACCEL", None)
    assert rel_err(out["1"][0], out["0"][0]).max() < 1e-14
    assert np.abs(out["1"][1] - out["0"][1]).max() < 1e-5


def test_numpy_stream_and_sequential_on_large_scene(oracle):
    from raytracer_amd.render import render_hip
    cam, st, mats, surfs, lights = _scene(400, 60, n=2, depth=2)
    img = render_hip(cam, st, mats, surfs, lights, 48, 27, jitter=("numpy", 5), verbose=False)
    ref = oracle.render(cam, st, mats, surfs, lights, 48, 27, jitter=("numpy", 5))
    assert rel_err(img, ref).max() < 1e-12
    img = render_hip(cam, st, mats, surfs, lights, 48, 27, jitter=("pixel", 5), sequential=True, verbose=False)
    ref = oracle.render(cam, st, mats, surfs, lights, 48, 27, jitter=("pixel", 5), sequential=True)
    assert rel_err(img, ref).max() < 1e-9


def test_regrouping_of_hits_changes_nothing():
    """rt_sort.hpp: the hits of level >= 1 are regrouped by the grid cell of the hit point before the
    shade kernel.  That is scheduling only: ray / hit / shadow-ray counts are identical and the images
    agree to the rounding of the framebuffer's accumulation order with the step switched off
    (RT_SORT_HITS=0)."""
    from raytracer_amd.render import render_hip
    cam, st, mats, surfs, lights = _scene(1500, 200, seed=4, n=2, depth=4)
    out = {}
    for mode in ("1", "0"):
        os.environ["RT_SORT_HITS"] = mode
        try:
            s64, s32 = {}, {}
            out[mode] = (render_hip(cam, st, mats, surfs, lights, 160, 90, jitter=("pixel", 9), stats=s64, verbose=False),
                         render_hip(cam, st, mats, surfs, lights, 160, 90, jitter=("pixel", 9), precision="fp32", stats=s32, verbose=False),
                         s64, s32)
        finally:
            os.environ.pop("RT_SORT_HITS", None)
    base, o = out["0"], out["1"]
    assert base[2]["rays"] > 160 * 90 * 1.5          # deeper levels exist
    for k in ("rays", "hits", "shadow_rays", "segments"):
        assert o[2][k] == base[2][k], k
    for k in ("rays", "hits"):
        assert o[3][k] == base[3][k], k
    assert rel_err(o[0], base[0]).max() < 1e-14
    assert np.abs(o[1] - base[1]).max() < 1e-5


def test_far_hits_keep_the_culling_conservative():
    """A camera just above the floor whose pixel row above the centre looks 1e-6 rad below the horizon:
    primary (and reflected) rays hit the infinite floor ~1e6 units away.  The fan culling of
    k_shade_fast is parametrised from the light end so that such hits neither lose precision nor turn
    every record into a candidate (rt_shade_fast.hpp).  Whatever it skips must not contribute: against
    the exact fp64 image the culled fp32 image is nowhere worse than the brute-force mode's (RT_ACCEL=0).
    (It is better on the far row: there the fp32 sphere roots of the brute-force loop cancel, |oc|^2 ~ 1e12,
    and report spurious crossings of spheres that the light-end test has already excluded.)"""
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip, camera_rays, nearest_hit
    from raytracer_amd.synth import synthetic_scene_text
    W, H = 192, 108
    step = 1.6 / W                                        # pixel pitch on the screen (screen_width 1.6)
    tan_pitch = (0.5 * step + 1.4e-6) / 1.4               # screen distance 1.4
    lines = synthetic_scene_text(700, 100, shadow_rays=3, max_recursions=3, seed=12).splitlines()
    ci = next(i for i, l in enumerate(lines) if l.startswith("cam "))
    lines[ci] = f"cam 0.0 0.3 -75.0 0.0 {0.3 - 75.0 * tan_pitch!r} 0.0 0.0 1.0 0.0 1.4 1.6"
    cam, st, objs = parse_scene_text("\n".join(lines) + "\n")
    mats, surfs, lights = separate_objects(objs)
    cam.setup(W / H)
    o, d = camera_rays(cam, W, H, H // 2 - 1, H // 2)
    t, idx, _ = nearest_hit(surfs, o, d)
    assert (idx >= 0).any() and t[idx >= 0].max() > 1e5   # the frame holds far floor hits
    out = {}
    for mode in ("1", "0"):
        os.environ["RT_ACCEL"] = mode
        try:
            out[mode] = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 3), precision="fp32", verbose=False)
        finally:
            os.environ.pop("RT_ACCEL", None)
    ref = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 3), verbose=False)
    err_accel, err_brute = np.abs(out["1"] - ref), np.abs(out["0"] - ref)
    assert (err_accel <= err_brute + 2e-6).all()
    assert np.abs(_q(out["1"]) - _q(ref)).max() <= 1
    far_row = H // 2 - 1
    print(f"far row: max |fp32 - fp64| culled {err_accel[far_row].max():.2e}, brute force {err_brute[far_row].max():.2e}")


def test_hierarchy_larger_than_the_lds_staging(oracle):
    """More than 988 nodes (LDS_NODES): the hierarchy is not staged in LDS, the walks read it from global memory
    (load_node's second branch).  RT_BVH_LEAF=1 gives every record its own leaf, so 1400 records make
    ~1600 nodes."""
    from raytracer_amd.render import render_hip
    cam, st, mats, surfs, lights = _scene(1200, 200, seed=6, n=2, depth=3)
    os.environ["RT_BVH_LEAF"] = "1"
    try:
        stats = {}
        img = render_hip(cam, st, mats, surfs, lights, 80, 45, jitter=("pixel", 4), stats=stats, verbose=False)
        img32 = render_hip(cam, st, mats, surfs, lights, 80, 45, jitter=("pixel", 4), precision="fp32", verbose=False)
    finally:
        os.environ.pop("RT_BVH_LEAF", None)
    ref, ost = oracle.render(cam, st, mats, surfs, lights, 80, 45, jitter=("pixel", 4), return_stats=True)
    for k in ("rays", "hits", "shadow_rays", "segments"):
        assert stats[k] == ost[k], k
    assert rel_err(img, ref).max() < 1e-12
    assert np.abs(_q(img32) - _q(ref)).max() <= 1
