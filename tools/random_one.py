#!/usr/bin/env python3
"""One randomised test scene: fp32 error vs the oracle (run with RT_SHADE=general to use k_shade_fast)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_random_scenes import random_scene_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
for arg in sys.argv[1:]:
    opaque = arg[0] == "T"; seed = int(arg[1:])
    cam, st, objs = parse_scene_text(random_scene_text(1000 * int(opaque) + seed, opaque))
    mats, surfs, lights = separate_objects(objs)
    W, H = 56, 40
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
    img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False)
    d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
    bad = np.argwhere(d8.max(axis=2) > 1)
    print(arg, "shade", os.environ.get("RT_SHADE", "ls"), "channels>1:", int((d8 > 1).sum()), "max", int(d8.max()), "first bad pixels", bad[:4].tolist())
    for y, x in bad[:2]:
        print("   ", img[y, x], ref[y, x])
