"""More than 128 primitives of which only a handful are spheres / boxes (the rest planes, which stay outside the
hierarchies): the nearest-hit tree then has a single leaf group (no inner node) or one inner node — the degenerate
shapes of the heap layout walked by k_trace_pool / nearest_heap."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_sph,n_box", [(3, 0), (0, 2), (5, 3), (9, 0), (8, 8), (1, 1)])
def test_many_planes_few_solids(oracle, n_sph, n_box):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.image import quantize_u8
    from raytracer_amd.render import render_hip
    rng = np.random.default_rng(3 + 10 * n_sph + n_box)
    lines = ["cam 0 2 8 0 0.5 0 0 1 0 1.4 2", "set 0.2 0.3 0.4 2 3",
             "mtl 0.7 0.7 0.7 0.4 0.4 0.4 0.2 0.2 0.2 12 0", "mtl 0.8 0.3 0.2 1 1 1 0.3 0.3 0.3 40 0.4"]
    for k in range(130):      # planes below the scene: > 128 primitives with a handful of spheres / boxes
        lines.append(f"pln 0 1 0 {-3.0 - 0.01 * k} 1")
    for k in range(n_sph):
        lines.append(f"sph {rng.uniform(-3, 3):.4f} {rng.uniform(-0.5, 1.5):.4f} {rng.uniform(-3, 2):.4f} {rng.uniform(0.4, 0.9):.4f} {1 + k % 2}")
    for k in range(n_box):
        lines.append(f"box {rng.uniform(-3, 3):.4f} {rng.uniform(-0.5, 1.0):.4f} {rng.uniform(-3, 2):.4f} {rng.uniform(0.6, 1.2):.4f} {1 + k % 2}")
    lines.append("lgt 3 6 4 1 1 1 1 0.8 1.0"); lines.append("lgt -4 5 2 0.7 0.7 0.9 1 0.6 0.5")
    cam, st, objs = parse_scene_text("\n".join(lines) + "\n")
    mats, surfs, lights = separate_objects(objs)
    assert len(surfs) > 128
    W, H = 96, 64
    ref, ostats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 2), return_stats=True)
    s = {}
    a = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp64", jitter=("pixel", 2), stats=s, verbose=False)
    b = render_hip(cam, st, mats, surfs, lights, W, H, precision="fp32", jitter=("pixel", 2), verbose=False)
    assert rel_err(a, ref).max() < 1e-9
    for k in ("rays", "hits", "shadow_rays", "segments"):
        assert s[k] == ostats[k], k
    assert np.abs(quantize_u8(b).astype(int) - quantize_u8(ref).astype(int)).max() <= 1
