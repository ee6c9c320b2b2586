import os, sys
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_random_scenes import random_scene_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
seed = int(sys.argv[1]); opaque = sys.argv[2] == "1"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (56, 40)
text = random_scene_text(1000 * int(opaque) + seed, opaque)
if os.environ.get("SHOW"): print(text)
cam, st, objs = parse_scene_text(text)
mats, surfs, lights = separate_objects(objs)
ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
s = {}
img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False, stats=s)
i64 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp64", verbose=False)
d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
print("lib", os.environ.get("RT_HIP_LIBRARY"), "channels>1", int((d8 > 1).sum()), "max", int(d8.max()), "retested", s.get("retested"), "hits", s.get("hits"), "fp64 max rel", float((np.abs(i64 - ref) / np.maximum(np.abs(ref), 1e-12)).max()))
ys, xs, cs = np.nonzero(d8 > 1)
for y, x, c in zip(ys, xs, cs):
    print("pixel", y, x, "channel", c, "fp32", img[y, x], "ref", ref[y, x], "diff*255", (img[y, x] - ref[y, x]) * 255)
