#!/usr/bin/env python3
"""Randomised large scenes (GPU box): the culling hierarchy against the library's own brute-force mode.

For every seed: fp64 with the hierarchy must equal fp64 brute force (RT_ACCEL=0) to the rounding of the
framebuffer's accumulation order with identical ray / hit / shadow-ray / segment counts, and the fp32
image with the hierarchy must agree with fp64 within one 8-bit level on >= 99.5 % of the channels.  Run
with the audit build (RT_HIP_LIBRARY=raytracer_amd/librt_hip_debug.so RT_DEBUG_STATS=65536) the library
also reports, per frame, every exact (fp64) crossing of a record by a shadow segment that the culling
dropped before the sample block: that count must be 0.  Usage: tools/stress_accel.py [first_seed] [n_seeds]"""
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
        if seed % 3 == 2:
            # wide flat scene (synthetic generator) with random lights: inside the slab, on the floor, huge radii
            from raytracer_amd.synth import synthetic_scene_text
            rng = np.random.default_rng(seed)
            lines = [l for l in synthetic_scene_text(int(rng.integers(300, 2500)), int(rng.integers(0, 300)), seed=seed,
                                                     shadow_rays=int(rng.integers(1, 4)), max_recursions=int(rng.integers(1, 4))).splitlines()
                     if not l.startswith("lgt ")]
            for _ in range(int(rng.integers(1, 4))):
                pos = [rng.uniform(-50, 50), rng.uniform(-0.9, 45), rng.uniform(-50, 50)]
                lines.append("lgt " + " ".join(repr(float(v)) for v in pos) + f" 0.6 0.5 0.7 1.0 {rng.uniform(0.3, 1):.3f} {float(rng.choice([0.0, 1.0, 4.0, 15.0]))}")
            if seed % 2:
                ci = next(i for i, l in enumerate(lines) if l.startswith("cam "))
                lines[ci] = f"cam {rng.uniform(-40, 40):.4f} {rng.uniform(0, 10):.4f} {rng.uniform(-40, 40):.4f} 0.0 3.0 0.0 0.0 1.0 0.0 1.0 1.8"
            text = "\n".join(lines) + "\n"
        else:
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
        from raytracer_amd.image import quantize_u8
        within = float((np.abs(quantize_u8(out32["1"][0]).astype(int) - quantize_u8(ref).astype(int)) <= 1).mean())
        # (`worse` is reported only: for hits ~1e3 units away the fp32 sphere roots of the sample block cancel, the
        # brute-force loop then sees spurious crossings that can hide a crossing it misses; whether the culling is
        # conservative is judged exactly by the audit build: RT_HIP_LIBRARY=.../librt_hip_debug.so RT_DEBUG_STATS=65536)
        ok = e64 < 1e-12 and same and within >= 0.995
        bad += not ok
        if not ok and os.environ.get("STRESS_DETAIL"):
            d = (ea - eb).max(axis=2)
            for y, x in zip(*np.nonzero(d > 2e-6)):
                print(f"   pixel ({y},{x}) culled {out32['1'][0][y, x]} brute {out32['0'][0][y, x]} fp64 {ref[y, x]}")
        print(f"seed {seed}: {len(surfs)} surfaces, fp64 max rel diff {e64:.1e}, counts equal {same}, fp32 within one 8-bit level of fp64 {within:.4f}, worst (culled - brute) error {worse:.1e}"
              f"{'' if ok else '   <-- FAIL'}", flush=True)
    print("failures:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
