#!/usr/bin/env python3
"""fp64 / fp32 against the oracle with the settings the other sweeps keep small: shadow grids up to 10x10 (several
sample chunks, ragged last chunk), recursion depth up to 10, up to 12 lights (beyond the light-space kernel's table),
up to 6 planes, shadow intensity exactly 0 / 1, zero specular, radius 1e-3 .. 20.  tools/settings_survey.py [n [first]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from test_gpu_random_scenes import random_scene_text
from test_gpu_many_spheres import many_spheres_text
from authored_survey import authored_scene_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def varied_text(seed):
    rng = np.random.default_rng(900000 + seed)
    kind = seed % 4
    if kind == 0: text = random_scene_text(21000 + seed, seed % 8 == 0)
    elif kind == 1: text = authored_scene_text(seed)
    elif kind == 2: text = many_spheres_text(2 + seed % 15, 22000 + seed, seed % 8 == 2)
    else: text = random_scene_text(61000 + seed, seed % 8 == 3, big=True)
    lines = text.strip().splitlines()
    n_mat = sum(l.startswith("mtl") for l in lines)
    out = []; deep = False
    for l in lines:
        if l.startswith("set "):
            f = l.split()
            f[4] = str(int(rng.choice([1, 2, 3] if kind == 3 else [1, 2, 3, 4, 5, 6, 7, 8, 10])))     # (the CPU oracle is brute force)
            f[5] = str(int(rng.choice([0, 1, 3] if kind == 3 else [0, 1, 2, 3, 6, 10])))
            if int(f[5]) >= 6: f[4] = str(min(int(f[4]), 3))        # deep trees: up to 2^depth rays per pixel in the CPU oracle
            deep = int(f[5]) >= 6
            l = " ".join(f)
        out.append(l)
    for _ in range(int(rng.choice([0, 0, 1, 3] if (kind == 3 or deep) else [0, 0, 1, 3, 9]))):                      # more lights
        pos = rng.uniform(-6, 6, 3); pos[1] = rng.uniform(0.5, 9)
        out.append("lgt " + " ".join(f"{v:.5f}" for v in pos) + f" {rng.uniform(0.1, 0.6):.3f} {rng.uniform(0.1, 0.6):.3f} {rng.uniform(0.1, 0.6):.3f}"
                   f" {float(rng.choice([0.0, 0.5, 1.0]))} {float(rng.choice([0.0, 0.4, 1.0]))} {float(rng.choice([0.0, 1e-3, 0.3, 2.0, 20.0]))}")
    for _ in range(int(rng.choice([0, 0, 1, 2, 4]))):                      # more planes
        nrm = rng.normal(size=3)
        out.append(f"pln {nrm[0]:.5f} {nrm[1]:.5f} {nrm[2]:.5f} {rng.uniform(-9, -4):.4f} {int(rng.integers(1, n_mat + 1))}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    import time
    bad = 0; worst = 0
    for seed in range(FIRST, FIRST + N):
        t0 = time.time()
        if os.environ.get("RT_DEBUG_STATS"): print("seed", seed, flush=True)
        cam, st, objs = parse_scene_text(varied_text(seed))
        mats, surfs, lights = separate_objects(objs)
        W, H = 48, 36
        ref, ost = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), return_stats=True)
        s64, s32 = {}, {}
        i64 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), stats=s64, verbose=False)
        i32 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", stats=s32, verbose=False)
        e64 = float((np.abs(i64 - ref) / np.maximum(np.abs(ref), 1e-3)).max())
        d8 = np.abs(quantize_u8(i32).astype(int) - quantize_u8(ref).astype(int))
        same = all(s64[k] == ost[k] for k in ("rays", "hits", "shadow_rays", "segments")) and all(s32[k] == ost[k] for k in ("rays", "hits", "shadow_rays"))
        worst = max(worst, int(d8.max()))
        if d8.max() > 1 or e64 > 1e-9 or not same:
            bad += 1
            print("seed", seed, "kind", seed % 4, "surfaces", len(surfs), "lights", len(lights), "n", st.root_number_shadow_rays, "depth", st.max_recursions,
                  "channels > 1:", int((d8 > 1).sum()), "max", int(d8.max()), "fp64 rel err", e64, "counts equal", same, flush=True)
        if time.time() - t0 > 5 or seed % 50 == 0: print(f"  (seed {seed}: {len(surfs)} surfaces, {len(lights)} lights, n {st.root_number_shadow_rays}, depth {st.max_recursions}: {time.time() - t0:.1f} s)", flush=True)
    print(f"{N} scenes, {bad} bad, worst fp32 diff {worst}")
