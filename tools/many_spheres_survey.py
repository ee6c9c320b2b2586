#!/usr/bin/env python3
"""fp32-vs-oracle sweep over random 7-16 sphere scenes (tests/test_gpu_many_spheres.py's generator): tools/many_spheres_survey.py [n_seeds [first_seed]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_many_spheres import many_spheres_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; worst = 0
for seed in range(FIRST, FIRST + N):
    n_sph = 7 + seed % 10
    cam, st, objs = parse_scene_text(many_spheres_text(n_sph, 5000 + seed, seed % 2 == 1))
    mats, surfs, lights = separate_objects(objs)
    W, H = 72, 48
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
    img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False)
    d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
    worst = max(worst, int(d8.max()))
    if d8.max() > 1:
        bad += 1; print("seed", seed, "spheres", n_sph, "channels > 1:", int((d8 > 1).sum()), "max", int(d8.max()))
print(f"{N} scenes, {bad} with a channel off by more than one level, worst diff {worst}")
