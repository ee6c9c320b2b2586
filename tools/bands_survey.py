#!/usr/bin/env python3
"""Host-output frames of >= 1 M pixels are rendered in row bands copied while the next band renders: banded output
(RT_BANDS = 3, 5, 16) against the unbanded frame (RT_BANDS=1) on randomised scenes, both precisions, all output
formats, stats summed.  tools/bands_survey.py [n [first]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from test_gpu_random_scenes import random_scene_text
from authored_survey import authored_scene_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import GpuScene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(FIRST, FIRST + N):
    kind = seed % 3
    text = random_scene_text(41000 + seed, seed % 2 == 0) if kind == 0 else (authored_scene_text(seed) if kind == 1 else random_scene_text(62000 + seed, seed % 4 == 0, big=True))
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    n, d = min(int(st.root_number_shadow_rays), 2), min(int(st.max_recursions), 3)
    W, H = [(1280, 832), (1031, 1019), (4096, 257)][seed % 3]
    errs = []
    with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs:
        for prec, fmt in (("fp32", "f32"), ("fp32", "u8"), ("fp32", "f64"), ("fp64", "f64"), ("fp64", "f32")):
            os.environ["RT_BANDS"] = "1"
            full = gs.render(cam, W, H, n, d, precision=prec, out_format=fmt, jitter=("pixel", seed)).copy(); s0 = dict(gs.last_stats)
            for nb in ("3", "5", "16"):
                os.environ["RT_BANDS"] = nb
                img = gs.render(cam, W, H, n, d, precision=prec, out_format=fmt, jitter=("pixel", seed)); s1 = dict(gs.last_stats)
                if fmt == "u8": ok = np.abs(img.astype(int) - full.astype(int)).max() <= (1 if prec == "fp32" else 0)
                else: ok = np.abs(img.astype(np.float64) - full.astype(np.float64)).max() <= (2e-6 if prec == "fp32" else 1e-12) * max(1.0, float(np.abs(full).max()))
                if not ok: errs.append(f"{prec}/{fmt} bands {nb}: image")
                if any(s0[k] != s1[k] for k in ("rays", "hits", "shadow_rays")): errs.append(f"{prec}/{fmt} bands {nb}: counters")
    os.environ.pop("RT_BANDS", None)
    if errs:
        bad += 1; print("seed", seed, "kind", kind, "size", (W, H), ":", "; ".join(errs[:6]), flush=True)
    print(f"  (seed {seed}: {len(surfs)} surfaces ok)" if not errs else "", flush=True)
print(f"{N} scenes, {bad} with a failure")
