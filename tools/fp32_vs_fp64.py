#!/usr/bin/env python3
"""GPU fp32 vs GPU fp64 of the same frame (the fp64 mode equals the oracle to 1e-12, tests/test_gpu_parity.py):
max 8-bit difference, channels off by more than one level, re-tested (hit, light) pairs, frame times.
    python tools/fp32_vs_fp64.py [scene[:n:depth] WxH]...      (default: a fixed list)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracer_amd import load_scene, parse_scene_text, separate_objects   # noqa: E402
from raytracer_amd.image import quantize_u8                                # noqa: E402
from raytracer_amd.render import GpuScene                                  # noqa: E402
from raytracer_amd.synth import synthetic_scene_text                       # noqa: E402

DEFAULT = ["pool.txt:5:3 960x540", "pool.txt:5:3 1920x1080", "pool.txt:5:3 3840x2160", "pool.txt:8:6 1920x1080", "original.txt 1920x1080",
           "boxes.txt 640x480", "synth_small.txt 480x270", "synth:4096:512:3:3 960x540"]


def load(spec):
    parts = spec.split(":")
    if parts[0] == "synth":
        ns, nb, n, d = (int(x) for x in parts[1:5])
        cam, st, objs = parse_scene_text(synthetic_scene_text(ns, nb, shadow_rays=n, max_recursions=d))
        mats, surfs, lights = separate_objects(objs)
        return cam, st, mats, surfs, lights
    n = int(parts[1]) if len(parts) > 1 else None
    d = int(parts[2]) if len(parts) > 2 else None
    return load_scene(os.path.join(ROOT, "scenes", parts[0]), n, d)


def main():
    args = sys.argv[1:]
    cases = [" ".join(args[i:i + 2]) for i in range(0, len(args), 2)] or DEFAULT
    for case in cases:
        spec, size = case.split()
        W, H = (int(x) for x in size.split("x"))
        cam, st, mats, surfs, lights = load(spec)
        cam.setup(W / H)
        n, d = int(st.root_number_shadow_rays), int(st.max_recursions)
        with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
            ref = gs.render(cam, W, H, n, d, precision="fp64", jitter=1234)
            img = gs.render(cam, W, H, n, d, precision="fp32", jitter=1234)
            s32 = dict(gs.last_stats)
            t0 = time.perf_counter()
            for _ in range(3):
                img = gs.render(cam, W, H, n, d, precision="fp32", jitter=1234)
            dt = (time.perf_counter() - t0) / 3
        d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
        pairs = s32["hits"] * len(lights)
        print(f"{case:34s} max8 {d8.max()}  >1: {(d8 > 1).sum():5d} of {d8.size:9d}  max|lin| {np.abs(img - ref).max():.2e}  "
              f"retested {s32['retested']:8d} = {s32['retested'] / max(pairs, 1):.2e} of (hit,light)  host-out {dt * 1e3:7.2f} ms", flush=True)
        if d8.max() > 1:
            ys, xs, cs = np.nonzero(d8 > 1)
            for y, x, c in list(zip(ys, xs, cs))[:5]:
                print(f"     pixel ({x},{y}) ch {c}: fp32 {img[y, x, c]:.6f} fp64 {ref[y, x, c]:.6f}")


if __name__ == "__main__":
    main()
