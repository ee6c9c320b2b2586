#!/usr/bin/env python3
"""pool.txt plus k extra spheres at 4K (does the headline kernel's speed transfer to slightly larger scenes?):
   tools/pool_plus.py [extra_spheres] [frames]   (RT_SHADE=g forces the general kernel for comparison)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracer_amd import _lib, parse_scene_text, separate_objects
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
text = open(os.path.join(ROOT, "scenes", "pool.txt")).read()
rng = np.random.default_rng(7)
sph = [l.split() for l in text.splitlines() if l.startswith("sph ")]
c = np.array([[float(v) for v in s_[1:4]] for s_ in sph]); r = float(sph[0][4])
for i in range(k):
    pos = c.mean(0) + np.array([rng.uniform(-1.5, 1.5), 0.0, rng.uniform(-1.5, 1.5)])
    text += f"sph {pos[0]:.4f} {pos[1]:.4f} {pos[2]:.4f} {r} {sph[i % len(sph)][5]}\n"
for i in range(int(os.environ.get("N_BOX", "0"))):      # extra boxes on the table ("box cx cy cz edge material")
    pos = c.mean(0) + np.array([rng.uniform(-2.5, 2.5), 0.0, rng.uniform(-2.5, 2.5)])
    text += f"box {pos[0]:.4f} {pos[1] - r + 0.4:.4f} {pos[2]:.4f} 0.8 {sph[i % len(sph)][5]}\n"
cam, st, objs = parse_scene_text(text)
mats, surfs, lights = separate_objects(objs)
W, H = 3840, 2160
cam.setup(W / H)
with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
    best = None
    for _ in range(frames):
        gs.render(cam, W, H, int(os.environ.get("N_SH", "5")), 3, precision="fp32", jitter=1234, flags=_lib.RT_FLAG_TIMING)
        s = gs.last_stats
        if best is None or s["total_ms"] < best["total_ms"]:
            best = dict(s)
    print(json.dumps({"spheres": len(sph) + k, "frame_ms": round(best["total_ms"], 3), "shade_ms": round(best["shade_ms"], 3),
                      "trace_ms": round(best["trace_ms"], 3), "retested": best["retested"]}))
