#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference).

Run only in the build container (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/gen_golden.py

The reference has no tests or golden vectors of its own, so these fixtures are what
pins the oracle (oracle/rt_oracle.c) and, through it, the HIP path.  Each fixture stores inputs,
the jitter mode/seed and the reference's float64 outputs.  Nothing of the reference's source is
stored — only numbers it computed.

Jitter: mode A = `np.random.seed(seed)` (the stock stream); mode B = `np.random.random` patched
with the pixel-keyed generator of raytracer_amd/jitter.py (SURVEY.md §8c "Jitter control").
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import ray_tracer as ref  # noqa: E402  (the reference)
from surfaces.sphere import Sphere as RefSphere  # noqa: E402
from surfaces.infinite_plane import InfinitePlane as RefPlane  # noqa: E402
from surfaces.cube import Cube as RefCube  # noqa: E402
from camera import Camera as RefCamera  # noqa: E402
from material import Material as RefMaterial  # noqa: E402

from raytracer_amd.jitter import PixelKeyedJitter  # noqa: E402

GOLD = os.path.join(HERE, "golden")
SEED = 1234


@contextlib.contextmanager
def pixel_keyed_numpy(seed):
    """Patch np.random.random so the reference consumes pixel-keyed jitter (mode B)."""
    jit = PixelKeyedJitter(seed)
    real = np.random.random
    state = {"axis": 0, "pair": None}

    def fake(M):
        f = sys._getframe(1)
        assert f.f_code.co_name == "compute_soft_shadow_batch", f.f_code.co_name
        outer = f.f_back
        assert outer.f_code.co_name == "render_vectorized", outer.f_code.co_name
        i, j, n = f.f_locals["i"], f.f_locals["j"], f.f_locals["n_shadow_rays"]
        light = f.f_locals["light"]
        lights = outer.f_locals["lights"]
        li = next(k for k, l in enumerate(lights) if l is light)
        pixels = outer.f_locals["hit_pixels"]
        depth = outer.f_locals["depth"]
        assert len(pixels) == M
        if state["axis"] == 0:
            state["pair"] = jit.uniforms(pixels, depth, li, i * n + j)
        u = state["pair"][state["axis"]]
        state["axis"] ^= 1
        return u

    np.random.random = fake
    try:
        yield
    finally:
        np.random.random = real


@contextlib.contextmanager
def pixel_keyed_scalar(seed):
    """Mode B for the reference's sequential render(): np.random.random() (scalar) is called twice per
    shadow-grid cell from compute_soft_shadow (ray_tracer.py:156-157); key = (pixel, level, light, cell)
    with level = max_depth - depth (the sequential renderer counts depth down)."""
    jit = PixelKeyedJitter(seed)
    real = np.random.random
    state = {"axis": 0, "pair": None}

    def fake():
        f = sys._getframe(1)
        assert f.f_code.co_name == "compute_soft_shadow", f.f_code.co_name
        cc = f.f_back
        assert cc.f_code.co_name == "compute_color", cc.f_code.co_name
        i, j, n = f.f_locals["i"], f.f_locals["j"], f.f_locals["n_shadow_rays"]
        light = f.f_locals["light"]
        li = next(k for k, l in enumerate(cc.f_locals["lights"]) if l is light)
        r = cc
        while r.f_code.co_name != "render":
            r = r.f_back
        pixel = r.f_locals["y"] * r.f_locals["width"] + r.f_locals["x"]
        level = r.f_locals["max_depth"] - cc.f_locals["depth"]
        if state["axis"] == 0:
            ui, uj = jit.uniforms(np.array([pixel]), level, li, i * n + j)
            state["pair"] = (float(ui[0]), float(uj[0]))
        u = state["pair"][state["axis"]]
        state["axis"] ^= 1
        return u

    np.random.random = fake
    try:
        yield
    finally:
        np.random.random = real


def ref_scene(path, n=None, depth=None):
    cam, st, objs = ref.parse_scene_file(path)
    if n is not None:
        st.root_number_shadow_rays = int(n)
    if depth is not None:
        st.max_recursions = int(depth)
    mats, surfs, lights = ref.separate_objects(objs)
    return cam, st, mats, surfs, lights


def ref_render(path, W, H, mode, seed, n=None, depth=None):
    cam, st, mats, surfs, lights = ref_scene(path, n, depth)
    log = io.StringIO()
    with contextlib.redirect_stdout(log):
        if mode == "A":
            np.random.seed(seed)
            img = ref.render_vectorized(cam, st, mats, surfs, lights, W, H)
        else:
            with pixel_keyed_numpy(seed):
                img = ref.render_vectorized(cam, st, mats, surfs, lights, W, H)
    rays = int(log.getvalue().rsplit("(", 1)[1].split()[0])
    return img, rays


def ref_render_sequential(path, W, H, mode, seed, n=None, depth=None):
    cam, st, mats, surfs, lights = ref_scene(path, n, depth)
    with contextlib.redirect_stdout(io.StringIO()):
        if mode == "A":
            np.random.seed(seed)
            return ref.render(cam, st, mats, surfs, lights, W, H)
        with pixel_keyed_scalar(seed):
            return ref.render(cam, st, mats, surfs, lights, W, H)


def save(name, **arrays):
    arrays["numpy_version"] = np.array(np.__version__)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------- G1-G3, G7: images ---
def gen_images():
    cases = [
        # name, scene file (in repo scenes/), W, H, n, depth
        ("g1_pool_64", "pool.txt", 64, 64, 5, 3),
        ("g2_original_96x54", "original.txt", 96, 54, None, None),
        ("g3_boxes_80x60", "boxes.txt", 80, 60, None, None),
        ("g7_synth_small_48x27", "synth_small.txt", 48, 27, None, None),
        ("g8_pool_n1_40", "pool.txt", 40, 40, 1, 2),      # n <= 1 branch (ray_tracer.py:184-196)
        ("g9_original_d0_48x27", "original.txt", 48, 27, 2, 0),  # depth cap 0: no children
    ]
    for name, scene, W, H, n, depth in cases:
        path = os.path.join(ROOT, "scenes", scene)
        out = {}
        for mode in ("A", "B"):
            img, rays = ref_render(path, W, H, mode, SEED, n, depth)
            out[f"image_{mode}"] = img
            out[f"rays_{mode}"] = np.array(rays)
        cam, st, *_ = ref_scene(path, n, depth)
        save(name, scene=np.array(scene), width=np.array(W), height=np.array(H),
             n_shadow=np.array(st.root_number_shadow_rays), max_depth=np.array(st.max_recursions),
             seed=np.array(SEED), **out)


# ------------------------------------------- GS: the sequential renderer ------
def gen_sequential():
    cases = [
        ("gs1_original_48x27", "original.txt", 48, 27, None, None),   # transparency, depth 5
        ("gs2_pool_32", "pool.txt", 32, 32, 3, 2),
        ("gs3_boxes_40x30", "boxes.txt", 40, 30, None, None),
        ("gs4_original_d1_40x24", "original.txt", 40, 24, 2, 1),      # depth cap hit by transparent surfaces
        ("gs5_original_n1_d0_32x18", "original.txt", 32, 18, 1, 0),   # one shadow ray, no recursion
    ]
    for name, scene, W, H, n, depth in cases:
        path = os.path.join(ROOT, "scenes", scene)
        out = {f"image_{mode}": ref_render_sequential(path, W, H, mode, SEED, n, depth) for mode in ("A", "B")}
        cam, st, *_ = ref_scene(path, n, depth)
        save(name, scene=np.array(scene), width=np.array(W), height=np.array(H),
             n_shadow=np.array(st.root_number_shadow_rays), max_depth=np.array(st.max_recursions),
             seed=np.array(SEED), **out)


# ---------------------------------------------------------------- G4: primitives + nearest hit ---
def edge_rays(rng):
    """Hand-built rays around a unit-ish primitive at the origin: grazing, inside, axis-parallel,
    starting on the surface (t ~ 1e-6), pointing away."""
    o, d = [], []

    def add(org, dr):
        dr = np.asarray(dr, float)
        o.append(np.asarray(org, float)); d.append(dr / np.linalg.norm(dr))

    for s in (-1, 1):
        for ax in range(3):
            e = np.zeros(3); e[ax] = s
            add(-3 * e, e)                 # head-on along an axis (two parallel axes for the box)
            add(3 * e, e)                  # pointing away
            add(np.zeros(3), e)            # from the centre (inside)
            add(0.999999 * e, e)           # just inside the surface
            add(1.0000005 * e, e)          # 5e-7 outside, leaving
            add(1.0000005 * e, -e)         # 5e-7 outside, entering (t below the 1e-6 threshold)
            add(1.000002 * e, -e)          # 2e-6 outside, entering (t above the threshold)
            t = np.zeros(3); t[(ax + 1) % 3] = 1.0
            add(-3 * t + e, t)             # grazing: tangent to a unit sphere / along a box face
            add(-3 * t + e * (1 - 1e-9), t)
            add(-3 * t + e * (1 + 1e-9), t)
            add(-3 * t + 0.5 * e, t + 1e-11 * e)   # |d_axis| below the 1e-10 parallel threshold
            add(-3 * t + 0.5 * e, t + 1e-9 * e)    # just above it
    for _ in range(64):                    # inside, random directions
        add(rng.uniform(-0.4, 0.4, 3), rng.normal(size=3))
    for _ in range(64):                    # through edges / corners of the unit cube
        c = rng.choice([-1.0, 1.0], 3) * np.array([1, 1, rng.uniform(-1, 1)])[rng.permutation(3)]
        start = c * 3 + rng.normal(scale=0.01, size=3)
        add(start, c - start)
    return np.array(o), np.array(d)


def gen_primitives():
    rng = np.random.default_rng(42)
    M = 4096
    org = rng.uniform(-4, 4, (M, 3))
    tgt = rng.uniform(-1.5, 1.5, (M, 3))
    dr = tgt - org
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    eo, ed = edge_rays(rng)
    org = np.concatenate([org, eo]); dr = np.concatenate([dr, ed])
    # unnormalised directions too: the sphere test keeps a = d.d (sphere.py:50)
    org = np.concatenate([org, org[:256]]); dr = np.concatenate([dr, dr[:256] * rng.uniform(0.3, 3, (256, 1))])

    prims = {
        "sphere_unit": RefSphere([0, 0, 0], 1.0, 1),
        "sphere_off": RefSphere([0.3, -0.2, 0.5], 0.75, 1),
        "plane_y": RefPlane([0, 1, 0], -1.0, 1),
        "plane_tilt": RefPlane([1, 2, -0.5], 0.25, 1),
        "box_unit": RefCube([0, 0, 0], 2.0, 1),
        "box_off": RefCube([0.3, -0.2, 0.5], 1.3, 1),
    }
    out = {"origins": org, "dirs": dr}
    for name, p in prims.items():
        t, n = p.intersect_batch(org, dr)
        out[f"{name}_t"] = t
        out[f"{name}_n"] = n
    save("g4_primitives", **out)

    # nearest hit with exact ties: duplicate sphere; box face coplanar with a plane (plane first
    # and plane second); rays from above and below.
    def tie_case(order):
        objs = {
            "pln": RefPlane([0, 1, 0], -1.0, 1),
            "box": RefCube([0, -0.5, 3], 1.0, 2),       # bottom face y = -1 lies in the plane
            "sph": RefSphere([2, 0, 0], 1.0, 3),
            "sph2": RefSphere([2, 0, 0], 1.0, 4),       # exact duplicate
            "box2": RefCube([-2, 0, 0], 1.0, 5),
        }
        return [objs[k] for k in order]

    M2 = 2048
    o2 = np.concatenate([rng.uniform(-4, 4, (M2, 3)) * [1, 0.2, 1] + [0, -3, 0],   # below the plane
                         rng.uniform(-4, 4, (M2, 3)) * [1, 0.5, 1] + [0, 4, 0]])   # above
    t2 = rng.uniform(-1, 1, (2 * M2, 3)) * [3, 1, 3] + [0, -0.5, 1.5]
    d2 = t2 - o2
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    out = {"origins": o2, "dirs": d2}
    for tag, order in (("a", ["pln", "box", "sph", "sph2", "box2"]), ("b", ["sph2", "box2", "box", "sph", "pln"])):
        t, idx, n = ref.find_nearest_intersection_batch(o2, d2, tie_case(order))
        out[f"order_{tag}"] = np.array(order)
        out[f"t_{tag}"] = t; out[f"idx_{tag}"] = idx; out[f"n_{tag}"] = n
    save("g4_nearest_ties", **out)


# ------------------------------------------------------------------------- G5: shadow walks ---
def gen_shadow_walks():
    rng = np.random.default_rng(7)
    mats = [RefMaterial([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 0.0),     # 1 opaque
            RefMaterial([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 0.5),     # 2 half transparent
            RefMaterial([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 0.9),     # 3 mostly transparent
            RefMaterial([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 1e-7)]    # 4 drops below EPS at once
    surfs = [RefSphere([0, 0, 0], 1.0, 2), RefSphere([3, 0, 0], 1.0, 3), RefCube([6, 0, 0], 1.5, 2),
             RefSphere([9, 0, 0], 0.8, 1), RefSphere([0, 4, 0], 1.0, 4), RefPlane([0, 1, 0], -3.0, 3)]
    # 12 thin transparent slabs along z: a ray down the z axis crosses 24 faces -> hits the 20-step cap
    for k in range(12):
        surfs.append(RefCube([0, 8, 3 + 1.0 * k], 0.5, 3))
    materials_array = {"transparency": np.array([m.transparency for m in mats])}
    M = 3000
    o = rng.uniform(-2, 2, (M, 3)) + [-3, 0, 0]
    tgt = np.stack([rng.uniform(2, 12, M), rng.uniform(-1.2, 1.2, M), rng.uniform(-1.2, 1.2, M)], 1)
    d = tgt - o
    dist = np.linalg.norm(d, axis=1)
    d /= dist[:, None]
    maxd = dist * rng.uniform(0.3, 1.2, M)
    # straight through everything on the x axis, toward the EPS material, down the slab stack
    extra_o = np.array([[-5, 0, 0], [-5, 0.2, 0.1], [0, 0, 0], [0, 9, 0], [0, 8, 0], [0.1, 8.05, 0], [0, 8, 20]], float)
    extra_d = np.array([[1, 0, 0], [1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 0, 1], [0, 0, -1]], float)
    extra_m = np.array([20, 20, 20, 20, 30, 30, 30], float)
    o = np.concatenate([o, extra_o]); d = np.concatenate([d, extra_d]); maxd = np.concatenate([maxd, extra_m])
    trans = ref.compute_shadow_transmission_batch(o, d, maxd, surfs, None, materials_array)
    save("g5_shadow_walks", origins=o, dirs=d, max_distances=maxd, transmission=trans,
         scene=np.array("see tests/golden_scenes.py: shadow_walk_scene()"))


# ---------------------------------------------------------------------------- G6: camera rays ---
def gen_camera():
    out = {}
    cases = {"pool": ([0, 10, -2], [0, -100, -4], [0, 1, 0], 1.4, 1.0, 48, 27),
             "tilted": ([1.5, 2.0, 8.0], [0.2, -0.3, 0.1], [0.1, 1, 0.05], 1.5, 2.0, 37, 23)}
    for name, (pos, look, upv, sd, sw, W, H) in cases.items():
        cam = RefCamera(pos, look, upv, sd, sw)
        cam.setup(W / H)
        o, d = cam.generate_all_rays(W, H)
        o2, d2 = cam.generate_rays_for_rows(W, H, 5, 11)
        out[f"{name}_params"] = np.array(list(pos) + list(look) + list(upv) + [sd, sw, W, H], float)
        out[f"{name}_basis"] = np.concatenate([cam.forward, cam.right, cam.up, [cam.screen_height]])
        out[f"{name}_origins"] = o; out[f"{name}_dirs"] = d
        out[f"{name}_rows_5_11_dirs"] = d2
    save("g6_camera", **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if sys.argv[1:] == ["sequential"]:
        gen_sequential()
        sys.exit(0)
    gen_primitives()
    gen_shadow_walks()
    gen_camera()
    gen_images()
    gen_sequential()
