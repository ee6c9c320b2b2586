import os, sys
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from authored_survey import authored_scene_text
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_random_scenes import random_scene_text
from raytracer_amd import parse_scene_text, separate_objects, _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
seed, y, x = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "authored"
text = authored_scene_text(seed) if kind == "authored" else random_scene_text(1000 * int(kind == "random1") + seed, kind == "random1")
cam, st, objs = parse_scene_text(text)
mats, surfs, lights = separate_objects(objs)
W, H = (80, 60) if kind == "authored" else (56, 40)
cam.setup(W / H)
n = int(st.root_number_shadow_rays)
tag = os.environ.get("TAG", "")
def both(ls, depth, flags=0):
    with GpuScene(pack_scene(mats, surfs, ls, st.background_color)) as gs:
        i64 = gs.render(cam, W, H, n, depth, precision="fp64", jitter=seed)
        i32 = gs.render(cam, W, H, n, depth, precision="fp32", jitter=seed, flags=flags)
        s = dict(gs.last_stats)
    return i32, i64, s
for depth in range(0, int(st.max_recursions) + 1):
    for flags, fn in ((0, "culled"), (_lib.RT_FLAG_BRUTE_FORCE, "brute")):
        i32, i64, s = both(lights, depth, flags)
        d = np.abs(i32 - i64)
        ys, xs = np.nonzero(d.max(axis=2) > 3e-3)
        print(tag, "depth", depth, fn, "pixel fp32", i32[y, x], "fp64", i64[y, x], "retested", s.get("retested"), "| pixels off by > 4e-3:", list(zip(ys.tolist(), xs.tolist()))[:6], flush=True)
for li in range(len(lights)):
    i32, i64, s = both([lights[li]], int(st.max_recursions))
    print(tag, "only light", li, "pixel fp32", i32[y, x], "fp64", i64[y, x], flush=True)
