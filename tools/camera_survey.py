#!/usr/bin/env python3
"""Frame sizes and cameras the other sweeps keep fixed: odd / tiny / tall / wide frames (1x1 .. 257x3), ragged row
blocks of any height over 2-7 parts, cameras looking straight down / up / along an axis, tilted up vectors, extreme
screen distances and widths.  fp64 + fp32 vs oracle, tiles vs frame.  tools/camera_survey.py [n [first]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from test_gpu_random_scenes import random_scene_text
from authored_survey import authored_scene_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import GpuScene
from raytracer_amd.packing import rows_in_part
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(FIRST, FIRST + N):
    rng = np.random.default_rng(400000 + seed)
    text = random_scene_text(31000 + seed, seed % 4 == 0, big=(seed % 6 == 5)) if seed % 2 == 0 else authored_scene_text(seed)
    lines = text.strip().splitlines()
    mode = int(rng.integers(0, 6))
    pos = rng.uniform(-5, 5, 3); pos[1] = rng.uniform(0.5, 7)
    if mode == 0: look, up = pos + np.array([0, -1.0, 0]), np.array([0, 0, -1.0])                 # straight down
    elif mode == 1: look, up = pos + np.array([0, 1.0, 0]), np.array([1.0, 0, 0])                 # straight up
    elif mode == 2: look, up = pos + np.array([0, 0, -1.0]), np.array([0.3, 1.0, 0.2])            # along -z, tilted up vector
    elif mode == 3: look, up = np.zeros(3), rng.normal(size=3)                                    # arbitrary up
    elif mode == 4: look, up = rng.uniform(-1, 1, 3), np.array([0, 1.0, 0])
    else: look, up = pos + np.array([1.0, 0, 0]), np.array([0, 1.0, 1e-3])
    sd = float(rng.choice([0.05, 0.5, 1.0, 3.0, 40.0])); sw = float(rng.choice([0.1, 1.0, 2.0, 8.0]))
    lines = [("cam " + " ".join(repr(float(v)) for v in list(pos) + list(look) + list(up)) + f" {sd} {sw}") if l.startswith("cam ") else l for l in lines]
    cam, st, objs = parse_scene_text("\n".join(lines) + "\n")
    mats, surfs, lights = separate_objects(objs)
    n, d = min(int(st.root_number_shadow_rays), 3), min(int(st.max_recursions), 3)
    W, H = [(1, 1), (1, 37), (257, 3), (7, 5), (64, 64), (33, 65), (100, 9), (8, 8), (9, 17), (130, 70)][int(rng.integers(0, 10))]
    errs = []
    st.root_number_shadow_rays, st.max_recursions = n, d
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
    if not np.isfinite(ref).all():
        continue                                   # degenerate camera basis (up parallel to the view direction): NaN in the reference too
    with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
        kw = dict(jitter=("pixel", seed))
        cam.forward = None
        f64 = gs.render(cam, W, H, n, d, precision="fp64", **kw).copy()
        f32 = gs.render(cam, W, H, n, d, precision="fp32", **kw).copy()
        if (np.abs(f64 - ref) / np.maximum(np.abs(ref), 1e-3)).max() > 1e-9: errs.append(f"fp64 vs oracle {(np.abs(f64 - ref) / np.maximum(np.abs(ref), 1e-3)).max():.2e}")
        d8 = np.abs(quantize_u8(f32).astype(int) - quantize_u8(ref).astype(int))
        if d8.max() > 1: errs.append(f"fp32 vs oracle: max {d8.max()}, {(d8 > 1).sum()} channels")
        br, parts = int(rng.integers(1, 12)), int(rng.integers(2, 8))
        for prec, full, tol in (("fp64", f64, 1e-12), ("fp32", f32, 2e-6)):
            asm = np.full_like(full, np.nan)
            for part in range(parts):
                rows = rows_in_part(H, br, part, parts)
                if len(rows) == 0:
                    continue
                asm[rows] = gs.render(cam, W, H, n, d, precision=prec, block_rows=br, part=part, n_parts=parts, **kw)
            if not (np.abs(asm - full).max() <= tol * max(1.0, np.abs(full).max())): errs.append(f"tiles {prec} ({br} rows x {parts} parts): {np.abs(asm - full).max():.2e}")
    if errs:
        bad += 1
        print("seed", seed, "mode", mode, "size", (W, H), "sd", sd, "sw", sw, "surfaces", len(surfs), ":", "; ".join(errs), flush=True)
    if seed % 50 == 0: print(f"  (seed {seed})", flush=True)
print(f"{N} scenes, {bad} with a failure")
