"""The reference's shadow walk multiplies at most 20 crossings (`for _ in range(20)`, ray_tracer.py:257): behind more
transparent surfaces than that a sample keeps the product of the FIRST 20, whatever follows — even an opaque blocker.
The fp32 kernels form the product over all crossings of a segment, so they have to notice segments that can cross more
than 20 surfaces and leave them to the exact walk (found by tools/stress_accel.py seed 349: 276 spheres of
transparency 0.9 / 1.0, 4 % of the channels off by up to 6 levels before the crossing count existed)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


def stack_text(n_shells, tau, extra, radius):
    """Floor, a light high above it, and between them `n_shells` thin transparent spheres lined up on the vertical
    through the origin (2 crossings each for a shadow segment from the floor below), then an OPAQUE sphere above
    the stack: with more than 10 shells the walk never reaches it.  `extra` unrelated small spheres far to the side
    push the scene past the LDS / hierarchy thresholds."""
    lines = ["cam 0 3 14 0 3 0 0 1 0 1.5 2", "set 0.1 0.15 0.2 2 2",
             "mtl 0.8 0.8 0.8 0.3 0.3 0.3 0.1 0.1 0.1 10 0",
             f"mtl 0.3 0.5 0.9 0.5 0.5 0.5 0.0 0.0 0.0 30 {tau}",
             "mtl 0.9 0.2 0.2 0.5 0.5 0.5 0.0 0.0 0.0 30 0",
             "pln 0 1 0 0 1"]
    for k in range(n_shells):
        lines.append(f"sph 0 {0.6 + 0.5 * k:.3f} 0 0.22 2")
    lines.append(f"sph 0 {0.6 + 0.5 * n_shells + 0.4:.3f} 0 0.5 3")
    rng = np.random.default_rng(n_shells * 7 + extra)
    for _ in range(extra):
        lines.append(f"sph {rng.uniform(6, 14):.4f} {rng.uniform(0.2, 6):.4f} {rng.uniform(-8, 2):.4f} {rng.uniform(0.05, 0.2):.4f} {int(rng.integers(1, 4))}")
    lines.append(f"lgt 0 40 0 1 1 1 1 0.9 {radius}")
    lines.append(f"lgt 6 9 5 0.5 0.5 0.4 1 0.6 {radius}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("n_shells,tau,extra,radius", [
    (9, 0.9, 0, 0.0),       # 18 crossings + the blocker: within the cap, light-space kernel
    (12, 0.9, 0, 0.0),      # 24 crossings: the opaque sphere above is never reached
    (12, 1.0, 0, 0.5),      # fully transparent shells use up the walk: the blocker casts no shadow at all
    (15, 0.9, 0, 1.0),
    (14, 0.95, 60, 0.3),    # <= 128 surfaces: general kernel, brute loops
    (14, 1.0, 300, 0.0),    # culling hierarchy
    (24, 0.9, 300, 0.8),
])
def test_more_crossings_than_the_walk_visits(oracle, n_shells, tau, extra, radius):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    cam, st, objs = parse_scene_text(stack_text(n_shells, tau, extra, radius))
    mats, surfs, lights = separate_objects(objs)
    W, H = 96, 72
    ref, ostats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 5), return_stats=True)
    s64, s32 = {}, {}
    img64 = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp64", jitter=("pixel", 5), stats=s64, verbose=False)
    img32 = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp32", jitter=("pixel", 5), stats=s32, verbose=False)
    assert rel_err(img64, ref).max() < 1e-9
    for k in ("rays", "hits", "shadow_rays"):
        assert s64[k] == ostats[k] and s32[k] == ostats[k], k
    d8 = np.abs(_q(img32) - _q(ref))
    print(f"{n_shells} shells tau {tau} + {extra}: fp32 max 8-bit diff {d8.max()}, {(d8 > 1).sum()} channels > 1, {s32['retested']} pairs re-tested")
    assert d8.max() <= 1


def test_stress_scene_349(oracle):
    """The scene that showed it (tools/stress_accel.py seed 349)."""
    from test_gpu_random_scenes import random_scene_text
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    cam, st, objs = parse_scene_text(random_scene_text(50349, False, big=True))
    mats, surfs, lights = separate_objects(objs)
    ref = oracle.render(cam, st, mats, surfs, lights, 96, 64, jitter=("pixel", 349))
    s32 = {}
    img32 = render_hip(cam, st, mats, surfs, lights, 96, 64, precision="fp32", jitter=("pixel", 349), stats=s32, verbose=False)
    d8 = np.abs(_q(img32) - _q(ref))
    print(f"fp32 max 8-bit diff {d8.max()}, {(d8 > 1).sum()} channels > 1, {s32['retested']} of {3 * s32['hits']} pairs re-tested")
    assert d8.max() <= 1


# --- crossings closer than 2e-6 to each other count once in the walk: coincident surfaces ---------------------------

def glass_blocks_text(tau, floor_tau, below):
    """Transparent cubes stacked and side by side (shared faces, exactly representable coordinates), standing on the
    floor plane (bottom faces in the plane), a duplicated sphere and a duplicated plane."""
    cam = "cam 0.3 -3.5 7 0.5 1.5 0 0 1 0 1.4 2" if below else "cam 2.5 3.5 8 0.5 1.2 0 0 1 0 1.4 2"
    lines = [cam, "set 0.2 0.25 0.3 3 3",
             f"mtl 0.8 0.8 0.8 0.3 0.3 0.3 0.1 0.1 0.1 10 {floor_tau}",
             f"mtl 0.3 0.6 0.9 0.6 0.6 0.6 0.05 0.05 0.05 40 {tau}",
             f"mtl 0.9 0.5 0.2 0.6 0.6 0.6 0.0 0.0 0.0 20 {tau}",
             "mtl 0.9 0.9 0.2 0.5 0.5 0.5 0.0 0.0 0.0 20 0",
             "pln 0 1 0 0 1", "pln 0 0 1 -6 1", "pln 0 0 1 -6 1"]
    for x in (0.0, 1.0):
        for y in (0.5, 1.5, 2.5):
            lines.append(f"box {x} {y} 0 1 {2 + int(x + y) % 2}")
    lines.append("box 2.25 0.25 0.5 0.5 2")             # smaller block against the side of the wall, on the floor
    lines.append("sph -2 1 0.5 0.8 2"); lines.append("sph -2 1 0.5 0.8 3")     # the same sphere twice
    lines.append("sph 0.5 3.6 -1.5 0.4 4")
    lines.append("lgt -6 2.2 1.5 1 1 1 1 0.9 0.6")      # low: segments run through the vertical shared faces
    lines.append("lgt 0.8 9 0.7 0.8 0.8 0.7 1 0.8 1.0")  # high: through the horizontal ones
    lines.append("lgt 5 -4 3 0.6 0.6 0.6 1 0.7 0.0")     # below the floor
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("tau,floor_tau,below", [(0.7, 0.0, False), (0.5, 0.6, False), (0.7, 0.5, True), (1.0, 0.3, True)])
def test_coincident_surfaces(oracle, tau, floor_tau, below):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    cam, st, objs = parse_scene_text(glass_blocks_text(tau, floor_tau, below))
    mats, surfs, lights = separate_objects(objs)
    W, H = 120, 90
    ref, ostats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 11), return_stats=True)
    s64, s32 = {}, {}
    img64 = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp64", jitter=("pixel", 11), stats=s64, verbose=False)
    img32 = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp32", jitter=("pixel", 11), stats=s32, verbose=False)
    assert rel_err(img64, ref).max() < 1e-9
    for k in ("rays", "hits", "shadow_rays"):
        assert s64[k] == ostats[k] and s32[k] == ostats[k], k
    d8 = np.abs(_q(img32) - _q(ref))
    print(f"tau {tau} floor {floor_tau} below {below}: fp32 max 8-bit diff {d8.max()}, {(d8 > 1).sum()} channels > 1, {s32['retested']} of {3 * s32['hits']} pairs re-tested")
    assert d8.max() <= 1


def test_many_tiles_and_boxes_on_a_floor(oracle):
    """Scene creation's search for coincident faces on a floor of 40 x 40 transparent tiles (every side face shared)
    plus 300 separate boxes standing on the floor plane (faces in a plane, nothing shared): correct image, and boxes
    that merely stand on the plane do not send their shadow pairs to the exact walk."""
    import time
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    rng = np.random.default_rng(3)
    head = ["cam 0 6 14 0 0.5 0 0 1 0 1.4 2", "set 0.2 0.25 0.3 2 2",
            "mtl 0.8 0.8 0.8 0.3 0.3 0.3 0.1 0.1 0.1 10 0", "mtl 0.3 0.6 0.9 0.6 0.6 0.6 0.0 0.0 0.0 40 0.6",
            "mtl 0.9 0.5 0.2 0.6 0.6 0.6 0.0 0.0 0.0 20 0.4"]
    tail = ["sph 0 2.5 0 1.0 2", "lgt -6 9 5 1 1 1 1 0.9 1.0", "lgt 5 7 -4 0.7 0.7 0.6 1 0.8 0.0"]
    tiles = head + ["pln 0 1 0 -1.5 1"] + [f"box {ix - 20 + 0.5} -0.25 {iz - 30 + 0.5} 1.0 {2 + (ix + iz) % 2}" for ix in range(40) for iz in range(40)] + tail
    apart = head + ["pln 0 1 0 0 1"] + [f"box {(k % 20) * 1.5 - 15} 0.25 {(k // 20) * 1.5 - 16} 0.5 {2 + k % 2}" for k in range(300)] + tail
    for name, lines in (("tiles", tiles), ("boxes apart", apart)):
        cam, st, objs = parse_scene_text("\n".join(lines) + "\n")
        mats, surfs, lights = separate_objects(objs)
        ref = oracle.render(cam, st, mats, surfs, lights, 96, 64, jitter=("pixel", 2))
        s32 = {}
        t0 = time.time()
        img = render_hip(cam, st, mats, surfs, lights, 96, 64, precision="fp32", jitter=("pixel", 2), stats=s32, verbose=False)
        dt = time.time() - t0
        d8 = np.abs(_q(img) - _q(ref))
        frac = s32["retested"] / (2 * s32["hits"])
        print(f"{name}: {len(surfs)} surfaces, fp32 max 8-bit diff {d8.max()}, {s32['retested']} of {2 * s32['hits']} pairs re-tested, scene + frame {dt:.2f} s")
        assert d8.max() <= 1 and dt < 5.0
        if name == "boxes apart":
            assert frac < 0.2, frac
