#!/usr/bin/env python3
"""Randomised large scenes (GPU box): the culling hierarchy against the library's own brute-force mode.

For every seed: fp64 with the hierarchy must equal fp64 brute force (RT_ACCEL=0) to the rounding of the
framebuffer's accumulation order with identical ray / hit / shadow-ray / segment counts, and the fp32
image with the hierarchy must be nowhere worse against fp64 than the fp32 brute-force image (the culling
may only skip records that contribute nothing).  Usage: tools/stress_accel.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from raytracer_amd import parse_scene_text, separate_objects   # noqa: E402
from raytracer_amd.render import render_hip                    # noqa: E402
from test_gpu_random_scenes import random_scene_text            # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0
    for seed in range(first, first + n):
        text = random_scene_text(50000 + seed, seed % 2 == 0, big=True)
        cam, st, objs = parse_scene_text(text)
        mats, surfs, lights = separate_objects(objs)
        args = (cam, st, mats, surfs, lights, 96, 64)
        kw = dict(jitter=("pixel", seed))
        out64, out32 = {}, {}
        for mode in ("1", "0"):
            os.environ["RT_ACCEL"] = mode
            s64, s32 = {}, {}
            out64[mode] = (render_hip(*args, stats=s64, verbose=False, **kw), s64)
            out32[mode] = (render_hip(*args, precision="fp32", stats=s32, verbose=False, **kw), s32)
        os.environ.pop("RT_ACCEL", None)
        ref = out64["0"][0]
        scale = np.maximum(np.abs(ref), 1e-3)
        e64 = (np.abs(out64["1"][0] - ref) / scale).max()
        same = all(out64["1"][1][k] == out64["0"][1][k] for k in ("rays", "hits", "shadow_rays", "segments"))
        ea, eb = np.abs(out32["1"][0] - ref), np.abs(out32["0"][0] - ref)
        worse = (ea - eb).max()
        ok = e64 < 1e-12 and same and worse <= 2e-6
        bad += not ok
        print(f"seed {seed}: {len(surfs)} surfaces, fp64 max rel diff {e64:.1e}, counts equal {same}, fp32 worst (culled - brute) error {worse:.1e}"
              f"{'' if ok else '   <-- FAIL'}", flush=True)
    print("failures:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
