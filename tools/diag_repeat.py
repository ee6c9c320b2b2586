import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import load_scene
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
cam, st, mats, surfs, lights = load_scene(os.path.join(ROOT, "scenes/pool.txt"), 5, 3)
W, H = 3840, 2160
cam.setup(W / H)
for depth in (0, 1, 3):
    with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
        a = gs.render(cam, W, H, 5, depth, precision="fp32", jitter=1234)
        b = gs.render(cam, W, H, 5, depth, precision="fp32", jitter=1234)
        r = gs.render(cam, W, H, 5, depth, precision="fp64", jitter=1234)
    d = np.abs(a - b).max(axis=2)
    print("depth", depth, "max", d.max(), "pixels > 1e-6:", int((d > 1e-6).sum()), "> 1e-5:", int((d > 1e-5).sum()))
    ys, xs = np.nonzero(d > 1e-5)
    for y, x in list(zip(ys, xs))[:5]:
        print("  ", y, x, a[y, x], b[y, x], r[y, x])
