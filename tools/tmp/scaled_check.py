import os, sys
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
sys.argv_saved = sys.argv; sys.argv = [sys.argv[0], "0"]
import importlib.util
spec = importlib.util.spec_from_file_location("scaled_survey_mod", os.path.join(ROOT, "tools", "scaled_survey.py"))
src = open(os.path.join(ROOT, "tools", "scaled_survey.py")).read().split("N = int(sys.argv[1])")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "scaled_survey.py")}; exec(compile(src, "scaled_survey_head", "exec"), ns)
scaled = ns["scaled"]; random_scene_text = ns["random_scene_text"]
sys.argv = sys.argv_saved
from raytracer_amd import parse_scene_text, separate_objects, _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
s, opaque, seed = float(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3])
text = scaled(random_scene_text(1000 * int(opaque) + seed, opaque), s)
print(text)
cam, st, objs = parse_scene_text(text)
mats, surfs, lights = separate_objects(objs)
W, H = 56, 40
cam.setup(W / H)
n = int(st.root_number_shadow_rays); D = int(st.max_recursions)
def run(ls, depth, flags=0):
    ref = oracle.render(cam, st.__class__(st.background_color, n, depth) if False else st, mats, surfs, ls, W, H, jitter=("pixel", seed))
    return ref
for depth in range(0, D + 1):
    st.max_recursions = depth
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
    with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
        for flags, fn in ((0, "culled"), (_lib.RT_FLAG_BRUTE_FORCE, "brute")):
            i32 = gs.render(cam, W, H, n, depth, precision="fp32", jitter=seed, flags=flags)
            stt = dict(gs.last_stats)
            d8 = np.abs(quantize_u8(i32).astype(int) - quantize_u8(ref).astype(int))
            ys, xs = np.nonzero(d8.max(axis=2) > 1)
            print("depth", depth, fn, "bad pixels", list(zip(ys.tolist(), xs.tolist()))[:5], "retested", stt.get("retested"), "hits", stt.get("hits"))
            for y, x in list(zip(ys.tolist(), xs.tolist()))[:3]:
                print("    pixel", y, x, "fp32", i32[y, x], "ref", ref[y, x], "diff*255", (i32[y, x] - ref[y, x]) * 255)
st.max_recursions = D
for li in range(len(lights)):
    ref = oracle.render(cam, st, mats, surfs, [lights[li]], W, H, jitter=("pixel", seed))
    with GpuScene(pack_scene(mats, surfs, [lights[li]], st.background_color)) as gs:
        i32 = gs.render(cam, W, H, n, D, precision="fp32", jitter=seed)
    d8 = np.abs(quantize_u8(i32).astype(int) - quantize_u8(ref).astype(int))
    ys, xs = np.nonzero(d8.max(axis=2) > 1)
    print("only light", li, "bad pixels", list(zip(ys.tolist(), xs.tolist()))[:5])
    for y, x in list(zip(ys.tolist(), xs.tolist()))[:3]:
        print("    pixel", y, x, "fp32", i32[y, x], "ref", ref[y, x], "diff*255", (i32[y, x] - ref[y, x]) * 255)
