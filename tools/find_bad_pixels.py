#!/usr/bin/env python3
"""Pixels of a frame whose fp32 render differs from the fp64 render by more than one 8-bit level, with the row
re-rendered alone (row sets) in both modes and at depth caps 0..D to see which bounce level the difference enters at.
    python tools/find_bad_pixels.py synth:4096:512:3:3 3840x2160"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from fp32_vs_fp64 import load
from raytracer_amd.image import quantize_u8
from raytracer_amd.render import GpuScene

spec, size = sys.argv[1], sys.argv[2]
W, H = (int(x) for x in size.split("x"))
cam, st, mats, surfs, lights = load(spec)
cam.setup(W / H)
n, d = int(st.root_number_shadow_rays), int(st.max_recursions)
with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
    a = gs.render(cam, W, H, n, d, precision="fp32", jitter=1234)
    b = gs.render(cam, W, H, n, d, precision="fp64", jitter=1234)
    d8 = np.abs(quantize_u8(a).astype(int) - quantize_u8(b).astype(int))
    ys, xs = np.nonzero(d8.max(axis=2) > 1)
    print(len(ys), "pixels differ by more than one level")
    for y, x in list(zip(ys, xs))[:6]:
        print(f"pixel ({x},{y}): fp32 {a[y, x]} fp64 {b[y, x]} diff {a[y, x] - b[y, x]}")
        for depth in range(d + 1):
            ra = gs.render(cam, W, H, n, depth, precision="fp32", jitter=1234, block_rows=1, part=int(y), n_parts=H)
            rb = gs.render(cam, W, H, n, depth, precision="fp64", jitter=1234, block_rows=1, part=int(y), n_parts=H)
            print(f"   depth cap {depth}: diff {ra[0, x] - rb[0, x]}")
