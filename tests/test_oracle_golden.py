"""Pins the CPU oracle (oracle/rt_oracle.c) against vectors computed by the reference itself
(tests/gen_golden.py).  CPU only.

Bars: sphere/box/camera/shadow-walk values bit-exact; plane-dependent values 1e-12 relative (the
reference's plane test uses BLAS np.dot); whole images 1e-9 relative per north_star
(observed ~4e-16: np.power vs libm pow).
"""
import numpy as np
import pytest

from conftest import load_golden, scene_path, rel_err
from golden_scenes import g4_primitives, g4_tie_surfaces, shadow_walk_scene
from raytracer_amd import Camera, load_scene

IMAGE_CASES = ["g1_pool_64", "g2_original_96x54", "g3_boxes_80x60", "g7_synth_small_48x27",
               "g8_pool_n1_40", "g9_original_d0_48x27"]


def test_mt19937_stream_matches_numpy(oracle):
    np.random.seed(1234)
    assert np.array_equal(np.random.random(5000), oracle.mt_random(1234, 5000))
    np.random.seed(0)
    assert np.array_equal(np.random.random(700), oracle.mt_random(0, 700))


def test_pixel_keyed_jitter_matches_numpy_definition(oracle):
    from raytracer_amd.jitter import PixelKeyedJitter
    pix = (np.arange(4096, dtype=np.uint64) * 2654435761 % (1 << 25)).astype(np.uint32)
    for depth, light, cell in [(0, 0, 0), (3, 4, 24), (6, 1, 63)]:
        a = PixelKeyedJitter(1234).uniforms(pix, depth, light, cell)
        b = oracle.jitter_cell(1234, pix, depth, light, cell)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[0].min() >= 0 and a[0].max() < 1 and abs(a[0].mean() - 0.5) < 0.03


@pytest.mark.parametrize("name", list(g4_primitives()))
def test_primitive_intersections(oracle, name):
    g = load_golden("g4_primitives")
    t, n = oracle.intersect(g4_primitives()[name], g["origins"], g["dirs"])
    t_ref, n_ref = g[name + "_t"], g[name + "_n"]
    assert np.array_equal(np.isinf(t), np.isinf(t_ref))
    if name == "plane_tilt":     # BLAS dot in the reference
        hit = np.isfinite(t_ref)
        assert rel_err(t[hit], t_ref[hit]).max() < 1e-12
        assert np.abs(n - n_ref).max() == 0
    else:
        assert np.array_equal(t, t_ref)
        assert np.array_equal(n, n_ref)
    assert np.isfinite(t_ref).sum() > 100 and np.isinf(t_ref).sum() > 100


@pytest.mark.parametrize("tag", ["a", "b"])
def test_nearest_hit_ties(oracle, tag):
    g = load_golden("g4_nearest_ties")
    surfs = g4_tie_surfaces(g["order_" + tag])
    t, idx, n = oracle.nearest_hit(surfs, g["origins"], g["dirs"])
    assert np.array_equal(idx, g["idx_" + tag])
    assert np.array_equal(t, g["t_" + tag])
    assert np.array_equal(n, g["n_" + tag])
    assert len(np.unique(idx)) >= 4


def test_shadow_walks(oracle):
    g = load_golden("g5_shadow_walks")
    mats, surfs = shadow_walk_scene()
    tr = oracle.shadow_transmission(mats, surfs, g["origins"], g["dirs"], g["max_distances"])
    assert np.array_equal(tr, g["transmission"])
    assert len(np.unique(np.round(g["transmission"], 12))) > 6      # several blocker counts present
    assert g["transmission"][-3] == pytest.approx(0.9 ** 20)        # the 20-step cap


@pytest.mark.parametrize("name", ["pool", "tilted"])
def test_camera_rays(oracle, name):
    g = load_golden("g6_camera")
    p = g[name + "_params"]
    W, H = int(p[11]), int(p[12])
    cam = Camera(p[0:3], p[3:6], p[6:9], p[9], p[10])
    cam.setup(W / H)
    basis = np.concatenate([cam.forward, cam.right, cam.up, [cam.screen_height]])
    assert np.array_equal(basis, g[name + "_basis"])
    o, d = oracle.camera_rays(cam, W, H)
    assert np.array_equal(o, g[name + "_origins"])
    assert np.array_equal(d, g[name + "_dirs"])
    _, d2 = oracle.camera_rays(cam, W, H, 5, 11)
    assert np.array_equal(d2, g[name + "_rows_5_11_dirs"])


@pytest.mark.parametrize("mode", ["A", "B"])
@pytest.mark.parametrize("name", IMAGE_CASES)
def test_images_match_reference(oracle, name, mode):
    g = load_golden(name)
    W, H = int(g["width"]), int(g["height"])
    cam, st, mats, surfs, lights = load_scene(scene_path(str(g["scene"])), int(g["n_shadow"]), int(g["max_depth"]))
    jitter = ("numpy" if mode == "A" else "pixel", int(g["seed"]))
    img, stats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=jitter, return_stats=True)
    ref = g["image_" + mode]
    assert img.shape == ref.shape == (H, W, 3)
    assert stats["rays"] == int(g["rays_" + mode])
    assert rel_err(img, ref).max() < 1e-9


def test_row_tiles_equal_full_frame(oracle):
    """Pixel-keyed jitter makes pixels independent: interleaved row tiles reassemble the frame."""
    from raytracer_amd.packing import rows_in_part
    W, H = 40, 30
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 2, 3)
    full = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 5))
    out = np.zeros_like(full)
    for part in range(3):
        tile = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", 5),
                             block_rows=4, part=part, n_parts=3)
        out[rows_in_part(H, 4, part, 3)] = tile
    assert np.array_equal(out, full)


# ------------------------------------------------- sequential renderer -------
SEQ_CASES = ["gs1_original_48x27", "gs2_pool_32", "gs3_boxes_40x30", "gs4_original_d1_40x24", "gs5_original_n1_d0_32x18"]


@pytest.mark.parametrize("name", SEQ_CASES)
@pytest.mark.parametrize("mode", ["A", "B"])
def test_sequential_oracle_matches_reference_render(oracle, name, mode):
    """oracle.render(sequential=True) vs images of the reference's own per-pixel `render`
    (ray_tracer.py:398-434): mode A = np.random.seed(1234) scalar stream, mode B = pixel-keyed.
    The scalar code's np.dot goes through BLAS -> tolerance 1e-9 relative, not bit-exact."""
    from raytracer_amd import load_scene
    g = load_golden(name)
    cam, st, mats, surfs, lights = load_scene(scene_path(str(g["scene"])), int(g["n_shadow"]), int(g["max_depth"]))
    jit = ("numpy", int(g["seed"])) if mode == "A" else ("pixel", int(g["seed"]))
    img = oracle.render(cam, st, mats, surfs, lights, int(g["width"]), int(g["height"]), jitter=jit, sequential=True)
    assert rel_err(img, g["image_" + mode]).max() < 1e-9


def test_sequential_and_vectorized_semantics_differ_where_expected(oracle):
    """At the depth cap the sequential renderer adds background*transparency (ray_tracer.py:370-377),
    the vectorized one nothing: the two oracles must differ on original.txt with max_depth 1."""
    from raytracer_amd import load_scene
    cam, st, mats, surfs, lights = load_scene(scene_path("original.txt"), 2, 1)
    a = oracle.render(cam, st, mats, surfs, lights, 40, 24, jitter=("pixel", 1234))
    b = oracle.render(cam, st, mats, surfs, lights, 40, 24, jitter=("pixel", 1234), sequential=True)
    assert np.abs(a - b).max() > 1e-3
