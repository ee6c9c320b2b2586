"""Randomised scenes (seeded) through the GPU path vs the CPU oracle.  The kernels skip work through
conservative claims (wave-level candidate sets, fully-lit / umbra shortcuts, culling hierarchy); a claim
that is not conservative shows up here as a differing image or a differing query count.  Scenes mix
spheres, planes and boxes, opaque and transparent materials, lights inside / touching / far from
objects, tiny and huge radii, cameras inside objects."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


def random_scene_text(seed, opaque, big=False):
    rng = np.random.default_rng(seed)
    n_mat = int(rng.integers(2, 6))
    lines = []
    cam = rng.uniform(-6, 6, 3); cam[1] = rng.uniform(0.5, 6)
    look = rng.uniform(-1.5, 1.5, 3)
    lines.append("cam " + " ".join(f"{v:.6f}" for v in list(cam) + list(look)) + f" 0 1 0 {rng.uniform(0.8, 2):.3f} {rng.uniform(1, 2.5):.3f}")
    lines.append(f"set {rng.uniform(0, 1):.3f} {rng.uniform(0, 1):.3f} {rng.uniform(0, 1):.3f} {int(rng.integers(1, 4))} {int(rng.integers(0, 5))}")
    for _ in range(n_mat):
        refl = rng.uniform(0, 0.6, 3) * (rng.random() < 0.6)
        tr = 0.0 if (opaque or rng.random() < 0.5) else float(rng.choice([0.3, 0.5, 0.9, 1e-7, 1.0]))
        shin = float(rng.choice([0, 1, 10, 50]))
        lines.append("mtl " + " ".join(f"{v:.4f}" for v in list(rng.uniform(0, 1, 6)) + list(refl)) + f" {shin} {tr}")
    n_sph, n_pln, n_box = int(rng.integers(0, 9)), int(rng.integers(0, 3)), int(rng.integers(0, 5))
    if seed % 7 == 3:
        n_box = 0                      # exercises the sphere-only small-scene kernels
    if seed % 5 == 1:
        n_sph += 9                     # beyond the small-scene limits -> general kernels
    if big:                            # > 128 surfaces -> culling hierarchy
        n_sph += int(rng.integers(100, 400)); n_box += int(rng.integers(30, 120))
    kinds = ["sph"] * n_sph + ["pln"] * n_pln + ["box"] * n_box
    rng.shuffle(kinds)
    for kd in kinds:
        m = int(rng.integers(1, n_mat + 1))
        if kd == "sph":
            c = rng.uniform(-4, 4, 3)
            r = float(rng.choice([0.01, 0.3, 0.8, 1.5, 6.0], p=[0.05, 0.3, 0.35, 0.25, 0.05]))
            lines.append(f"sph {c[0]:.5f} {c[1]:.5f} {c[2]:.5f} {r} {m}")
        elif kd == "pln":
            nrm = rng.normal(size=3); nrm[1] += 2.0
            lines.append(f"pln {nrm[0]:.5f} {nrm[1]:.5f} {nrm[2]:.5f} {rng.uniform(-3, -0.5):.4f} {m}")
        else:
            c = rng.uniform(-4, 4, 3)
            lines.append(f"box {c[0]:.5f} {c[1]:.5f} {c[2]:.5f} {rng.uniform(0.3, 2.5):.4f} {m}")
    for _ in range(int(rng.integers(1, 4))):
        pos = rng.uniform(-5, 5, 3); pos[1] = rng.uniform(-0.5, 8)
        lines.append("lgt " + " ".join(f"{v:.5f}" for v in pos) + f" {rng.uniform(0.2, 1):.3f} {rng.uniform(0.2, 1):.3f} {rng.uniform(0.2, 1):.3f}"
                     f" {rng.uniform(0, 1):.3f} {rng.uniform(0, 1):.3f} {float(rng.choice([0.0, 0.5, 1.5, 4.0]))}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("opaque", [True, False])
@pytest.mark.parametrize("seed", range(96))
def test_random_scene_fp64_exact_and_fp32_close(oracle, seed, opaque):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    text = random_scene_text(1000 * int(opaque) + seed, opaque)
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    W, H = 56, 40
    stats = {}
    img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), stats=stats, verbose=False)
    ref, ost = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), return_stats=True)
    for k in ("rays", "hits", "shadow_rays", "segments"):
        assert stats[k] == ost[k], (k, stats[k], ost[k], text)
    assert rel_err(img, ref).max() < 1e-9, text
    img32 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False)
    d8 = np.abs(_q(img32) - _q(ref))
    assert d8.max() <= 1, (int(d8.max()), int((d8 > 1).sum()), text)


@pytest.mark.parametrize("seed", range(6))
def test_random_scene_sequential_and_numpy_stream(oracle, seed):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    cam, st, objs = parse_scene_text(random_scene_text(5000 + seed, seed % 2 == 0))
    mats, surfs, lights = separate_objects(objs)
    W, H = 40, 28
    img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), sequential=True, verbose=False)
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), sequential=True)
    assert rel_err(img, ref).max() < 1e-9
    img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("numpy", seed), verbose=False)
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("numpy", seed))
    assert rel_err(img, ref).max() < 1e-9


@pytest.mark.parametrize("seed", range(16))
def test_random_large_scene_through_the_hierarchy(oracle, seed):
    """> 128 surfaces: nearest hits and shadow walks go through the culling hierarchy; overlapping and
    nested primitives, lights inside primitives, transparent materials."""
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.render import render_hip
    text = random_scene_text(7000 + seed, seed % 2 == 0, big=True)
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    assert len(surfs) > 128
    W, H = 48, 32
    stats = {}
    img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), stats=stats, verbose=False)
    ref, ost = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), return_stats=True)
    for k in ("rays", "hits", "shadow_rays", "segments"):
        assert stats[k] == ost[k], (k, stats[k], ost[k])
    assert rel_err(img, ref).max() < 1e-9
    img32 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False)
    d8 = np.abs(_q(img32) - _q(ref))
    assert d8.max() <= 1, (int(d8.max()), int((d8 > 1).sum()))
