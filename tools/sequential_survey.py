#!/usr/bin/env python3
"""Sequential-renderer semantics (RT_FLAG_SEQUENTIAL): fp64 and fp32 against the oracle's literal recursion over the
randomised small, authored and 7-16 sphere scenes.  tools/sequential_survey.py [n_seeds [first_seed]]"""
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
bad = 0; worst = 0; worst64 = 0.0
for seed in range(FIRST, FIRST + N):
    kind = seed % 3
    text = random_scene_text(7000 + seed, seed % 2 == 0) if kind == 0 else (authored_scene_text(seed) if kind == 1 else many_spheres_text(7 + seed % 10, 9000 + seed, seed % 2 == 1))
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    W, H = 64, 48
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), sequential=True)
    i64 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), sequential=True, verbose=False)
    i32 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), sequential=True, precision="fp32", verbose=False)
    e64 = float((np.abs(i64 - ref) / np.maximum(np.abs(ref), 1e-3)).max())
    d8 = np.abs(quantize_u8(i32).astype(int) - quantize_u8(ref).astype(int))
    worst = max(worst, int(d8.max())); worst64 = max(worst64, e64)
    if d8.max() > 1 or e64 > 1e-9:
        bad += 1; print("seed", seed, "kind", kind, "surfaces", len(surfs), "channels > 1:", int((d8 > 1).sum()), "max", int(d8.max()), "fp64 rel err", e64, flush=True)
print(f"{N} scenes, {bad} bad, worst fp32 diff {worst}, worst fp64 rel err {worst64:.2e}")
