#!/usr/bin/env python3
"""fp32-vs-oracle survey over the randomised test scenes: prints the scenes with the most channels
off by more than one 8-bit level (finds configurations the fp32 kernels get wrong)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_random_scenes import random_scene_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
rows = []
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0        # tools/random_fp32_survey.py [n_seeds [first_seed [W H]]]
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (56, 40)
for opaque in (True, False):
    for seed in range(FIRST, FIRST + N):
        text = random_scene_text(1000 * int(opaque) + seed, opaque)
        cam, st, objs = parse_scene_text(text)
        mats, surfs, lights = separate_objects(objs)
        ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
        img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False)
        d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
        rows.append(((d8 > 1).sum(), int(d8.max()), opaque, seed, len(surfs)))
rows.sort(reverse=True)
print("scenes:", len(rows), "with any channel > 1 level:", sum(r[0] > 0 for r in rows), "total channels > 1:", sum(r[0] for r in rows), "of", len(rows) * W * H * 3)
for r in rows[:12]:
    print("channels>1: %d max diff %d opaque %s seed %d surfaces %d" % r)
