"""GPU edge cases (through the C-ABI), each checked against the CPU oracle: empty and degenerate
scenes, ragged image sizes, queue overflow + automatic regrowth, lights behind the shaded surface
(own-plane lower bound path of the fp32 kernel), scenes outside the small-scene kernel's limits."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _parse(text):
    from raytracer_amd import parse_scene_text, separate_objects
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    return cam, st, mats, surfs, lights


def _both(text, W, H, oracle, precision="fp64", jitter=("pixel", 3)):
    from raytracer_amd.render import render_hip
    cam, st, mats, surfs, lights = _parse(text)
    stats = {}
    img = render_hip(cam, st, mats, surfs, lights, W, H, precision=precision, jitter=jitter, stats=stats, verbose=False)
    ref, ostats = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=jitter, return_stats=True)
    return img, ref, stats, ostats


def _q(img):
    from raytracer_amd.image import quantize_u8
    return quantize_u8(img).astype(int)


def test_empty_scene_is_background(oracle):
    text = "cam 0 0 5 0 0 0 0 1 0 1 1\nset 0.25 0.5 0.75 3 2\n"
    for prec in ("fp64", "fp32"):
        img, ref, stats, ostats = _both(text, 7, 5, oracle, prec)
        assert img.shape == (5, 7, 3)
        assert np.array_equal(img, ref) and np.allclose(img, [0.25, 0.5, 0.75])
        assert stats["rays"] == 35 and stats["hits"] == 0


def test_no_lights_and_zero_shadow_rays(oracle):
    base = "cam 0 0 5 0 0 0 0 1 0 1 1\nmtl 0.8 0.2 0.2 1 1 1 0.3 0.3 0.3 20 0\nsph 0 0 0 1 1\n"
    img, ref, *_ = _both(base + "set 0.1 0.1 0.1 4 2\n", 16, 16, oracle)            # no lights: black sphere
    assert rel_err(img, ref).max() < 1e-12
    text = base + "set 0.1 0.1 0.1 0 2\nlgt 3 3 3 1 1 1 1 0.8 1\n"                  # n_shadow = 0 -> one ray to the centre
    for prec, tol in (("fp64", 1e-12), ("fp32", None)):
        img, ref, stats, ostats = _both(text, 16, 16, oracle, prec)
        if tol:
            assert rel_err(img, ref).max() < tol and stats["shadow_rays"] == ostats["shadow_rays"]
        else:
            assert np.abs(_q(img) - _q(ref)).max() <= 1


@pytest.mark.parametrize("W,H", [(1, 1), (3, 2), (65, 33), (257, 3)])
def test_ragged_sizes(oracle, W, H):
    text = open("scenes/original.txt").read()
    img, ref, stats, ostats = _both(text, W, H, oracle)
    assert rel_err(img, ref).max() < 1e-12 and stats["rays"] == ostats["rays"]
    img32, ref, *_ = _both(text, W, H, oracle, "fp32")
    assert np.abs(_q(img32) - _q(ref)).max() <= 1


def test_queue_overflow_regrows_and_matches_oracle(oracle):
    """Nested transparent + reflective spheres around the camera: every hit spawns two children,
    a level holds more than 2 rays per pixel and the first attempt overflows the bounce queues."""
    text = ("cam 0 0 0 0 0 -1 0 1 0 1 2\nset 0.2 0.3 0.4 1 6\n"
            "mtl 0.5 0.5 0.5 0.5 0.5 0.5 0.4 0.4 0.4 10 0.5\n"
            "sph 0 0 0 2 1\nsph 0 0 0 3 1\nsph 0 0 0 4 1\nsph 0 0 0 5 1\n"
            "lgt 0 0 0 1 1 1 1 0.5 1\n")
    img, ref, stats, ostats = _both(text, 48, 32, oracle)
    assert max(stats["level_rays"]) > 2 * 48 * 32 + 1024            # would not fit the default queues
    assert stats["rays"] == ostats["rays"] and stats["hits"] == ostats["hits"]
    assert rel_err(img, ref).max() < 1e-9
    img32, ref, *_ = _both(text, 48, 32, oracle, "fp32")
    assert np.abs(_q(img32) - _q(ref)).max() <= 1


def test_light_below_the_floor_and_inside_a_sphere(oracle):
    """Lights behind the shaded surface: exercises the exact lower bound of the own-plane test
    (PLANE_LO path of k_shade_ls) and origins inside spheres."""
    text = ("cam 0 3 8 0 0 0 0 1 0 1.5 2\nset 0.2 0.2 0.3 3 2\n"
            "mtl 0.6 0.6 0.6 0.9 0.9 0.9 0.2 0.2 0.2 8 0\nmtl 0.2 0.5 0.9 1 1 1 0.1 0.1 0.1 40 0.6\n"
            "pln 0 1 0 -1 1\nsph 0 0.2 0 1.2 2\nsph 2.5 -0.2 0.5 0.8 1\n"
            "lgt 1 -3 2 0.9 0.9 0.9 1 0.7 1.5\nlgt 0 0.2 0 0.8 0.8 0.6 1 0.6 0.5\nlgt 3 4 3 0.5 0.5 0.5 1 0.9 1\n")
    img, ref, *_ = _both(text, 96, 64, oracle)
    assert rel_err(img, ref).max() < 1e-9
    img32, ref, *_ = _both(text, 96, 64, oracle, "fp32")
    d8 = np.abs(_q(img32) - _q(ref))
    print(f"light-behind-surface scene fp32: max 8-bit diff {d8.max()}, channels > 1 level {(d8 > 1).sum()} of {d8.size}")
    assert d8.max() <= 1


def test_scene_outside_small_kernel_limits_uses_general_kernel(oracle):
    """20 lights, 3 planes, 9 spheres, transparency: k_shade_fast path."""
    rng = np.random.default_rng(5)
    lines = ["cam 0 4 10 0 0 0 0 1 0 1.5 2", "set 0.3 0.4 0.5 2 2",
             "mtl 0.7 0.7 0.7 0.5 0.5 0.5 0.1 0.1 0.1 15 0", "mtl 0.2 0.6 0.3 1 1 1 0 0 0 50 0.4",
             "pln 0 1 0 -1 1", "pln 0 0 1 -6 1", "pln 1 0 0 -7 1"]
    for _ in range(9):
        c = rng.uniform([-3, -0.5, -3], [3, 1.5, 3])
        lines.append(f"sph {c[0]} {c[1]} {c[2]} {rng.uniform(0.3, 0.8)} {rng.integers(1, 3)}")
    for _ in range(20):
        c = rng.uniform([-5, 3, -2], [5, 7, 6])
        lines.append(f"lgt {c[0]} {c[1]} {c[2]} 0.1 0.1 0.1 1 0.8 0.8")
    text = "\n".join(lines) + "\n"
    img, ref, *_ = _both(text, 64, 40, oracle)
    assert rel_err(img, ref).max() < 1e-9
    img32, ref, *_ = _both(text, 64, 40, oracle, "fp32")
    d8 = np.abs(_q(img32) - _q(ref))
    assert d8.max() <= 1


def test_numpy_stream_jitter_on_overflow_scene(oracle):
    """Depth-first batches with both child kinds at every hit (stream order of T/R batches)."""
    text = ("cam 0 0 0 0 0 -1 0 1 0 1 2\nset 0.2 0.3 0.4 2 4\n"
            "mtl 0.5 0.5 0.5 0.5 0.5 0.5 0.4 0.4 0.4 10 0.5\nsph 0 0 0 2 1\nsph 0 0 0 3 1\n"
            "lgt 0.5 0 0 1 1 1 1 0.5 1\nlgt -1 0.5 0 0.5 0.5 0.5 1 0.8 0.5\n")
    img, ref, stats, ostats = _both(text, 24, 16, oracle, "fp64", ("numpy", 77))
    assert stats["batches"] == ostats["batches"] and stats["rays"] == ostats["rays"]
    assert rel_err(img, ref).max() < 1e-12


def test_many_async_frames_and_two_scenes_in_flight(oracle):
    """RT_FLAG_ASYNC: 60 frames enqueued without a sync (the library folds finished frames itself when
    48 are pending), alternating between two scene handles with their own streams; every frame's stats
    and both final images must equal the blocking render (to the rounding of the atomic accumulation order)."""
    import ctypes
    from raytracer_amd import _lib, load_scene
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.render import GpuScene
    cam, st, mats, surfs, lights = load_scene("scenes/original.txt", 2, 3)
    W, H = 160, 90
    cam.setup(W / H)
    packed = pack_scene(mats, surfs, lights, st.background_color)
    hip = ctypes.CDLL("libamdhip64.so")
    bufs = [ctypes.c_void_p(), ctypes.c_void_p()]
    for b in bufs:
        assert hip.hipMalloc(ctypes.byref(b), W * H * 3 * 4) == 0
    try:
        with GpuScene(packed) as a, GpuScene(packed) as b:
            ref = a.render(cam, W, H, 2, 3, precision="fp32", jitter=5)
            ref_stats = dict(a.last_stats)
            stats = [_lib.Stats() for _ in range(60)]
            for k, s in enumerate(stats):
                (a if k % 2 == 0 else b).render(cam, W, H, 2, 3, precision="fp32", jitter=5, out_ptr=bufs[k % 2].value,
                                                flags=_lib.RT_FLAG_ASYNC, stats=s)
            a.sync(); b.sync()
            for s in stats:
                assert (s.rays, s.hits, s.shadow_rays) == (ref_stats["rays"], ref_stats["hits"], ref_stats["shadow_rays"])
            for buf in bufs:
                out = np.empty((H, W, 3), dtype=np.float32)
                assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), buf, out.nbytes, 2) == 0   # device -> host
                assert np.abs(out - ref).max() < 2e-6      # float atomics: accumulation order of a pixel's child rays
    finally:
        for b in bufs:
            hip.hipFree(b)
