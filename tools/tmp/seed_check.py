import os, sys
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from authored_survey import authored_scene_text
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
seed = int(sys.argv[1])
text = authored_scene_text(seed)
print(text)
cam, st, objs = parse_scene_text(text)
mats, surfs, lights = separate_objects(objs)
W, H = 80, 60
ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
img64 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), verbose=False)
img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", verbose=False)
d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
print("lib", os.environ.get("RT_HIP_LIBRARY"), "channels>1", int((d8 > 1).sum()), "max", int(d8.max()))
ys, xs, cs = np.nonzero(d8 > 1)
for y, x, c in zip(ys, xs, cs):
    print("pixel", y, x, "channel", c, "fp32", img[y, x], "fp64", img64[y, x], "ref", ref[y, x])
