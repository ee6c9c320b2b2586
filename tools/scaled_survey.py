#!/usr/bin/env python3
"""The randomised test scenes with every length multiplied by a factor (0.01 ... 1000): the reference's epsilons are
absolute (1e-6 origin shift, t > 1e-6), the fp32 kernels' bounds mostly relative — scenes far from unit size find the
bounds that are neither.  fp32 against the oracle (<= 1 level) and fp64 against the oracle (<= 1e-9).
    python tools/scaled_survey.py [n_seeds [first_seed [random|authored|many|settings]]]
(authored: the grid-aligned scenes of tools/authored_survey.py — at factors that are not powers of two their exact
coincidences become coincidences within a few ulp.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_random_scenes import random_scene_text
sys.path.insert(0, os.path.join(ROOT, "tools"))
from authored_survey import authored_scene_text
from test_gpu_many_spheres import many_spheres_text
from settings_survey import varied_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)


def scaled(text, s):
    out = []
    for ln in text.splitlines():
        f = ln.split()
        if not f or f[0].startswith("#"):
            out.append(ln); continue
        v = [float(x) for x in f[1:]]
        if f[0] == "cam":
            v[0:6] = [x * s for x in v[0:6]]
        elif f[0] == "pln":
            v[3] *= s
        elif f[0] in ("sph", "box"):
            v[0:4] = [x * s for x in v[0:4]]
        elif f[0] == "lgt":
            v[0:3] = [x * s for x in v[0:3]]; v[8] *= s
        else:
            out.append(ln); continue
        if f[0] in ("pln", "sph", "box"):
            out.append(f[0] + " " + " ".join(repr(x) for x in v[:-1]) + f" {int(v[-1])}")
        else:
            out.append(f[0] + " " + " ".join(repr(x) for x in v))
    return "\n".join(out) + "\n"


N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
KIND = sys.argv[3] if len(sys.argv) > 3 else "random"
SEQ = os.environ.get("SEQ") == "1"       # SEQ=1: the sequential renderer's semantics (RT_FLAG_SEQUENTIAL)
W, H = {"random": (56, 40), "authored": (80, 60), "many": (72, 48)}.get(KIND, (48, 36))
bad = 0; worst = 0; worst64 = 0.0
SCALES = tuple(float(x) for x in os.environ["SCALES"].split(",")) if os.environ.get("SCALES") else (0.01, 0.1, 10.0, 100.0, 1000.0)
for s in SCALES:
    for opaque in ((True, False) if KIND == "random" else (None,)):
        for seed in range(FIRST, FIRST + N):
            base = (random_scene_text(1000 * int(opaque) + seed, opaque) if KIND == "random" else authored_scene_text(seed) if KIND == "authored"
                    else many_spheres_text(7 + seed % 10, 5000 + seed, seed % 2 == 1) if KIND == "many"      # the 7-16 sphere scenes
                    else varied_text(seed))      # "settings": large shadow grids, deep recursion, many lights / planes, large scenes
            text = scaled(base, s)
            cam, st, objs = parse_scene_text(text)
            mats, surfs, lights = separate_objects(objs)
            ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), sequential=SEQ)
            img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False, sequential=SEQ)
            i64 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp64", verbose=False, sequential=SEQ)
            d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
            e64 = float((np.abs(i64 - ref) / np.maximum(np.abs(ref), 1e-3)).max())
            worst = max(worst, int(d8.max())); worst64 = max(worst64, e64)
            if d8.max() > 1 or e64 > 1e-9:
                bad += 1
                print(f"scale {s} opaque {opaque} seed {seed}: {int((d8 > 1).sum())} channels > 1 level (max {int(d8.max())}), fp64 rel err {e64:.2e}, {len(surfs)} surfaces", flush=True)
    print(f"  (scale {s} done)", flush=True)
print(f"{(2 if KIND == 'random' else 1) * len(SCALES) * N} scenes, {bad} bad, worst fp32 diff {worst}, worst fp64 rel err {worst64:.2e}")
