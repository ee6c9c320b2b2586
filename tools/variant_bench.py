#!/usr/bin/env python3
"""k_shade_ls vs k_shade_fast (RT_SHADE=general) on scenes at the edges of the small-scene kernel's
range: pool.txt + 2 spheres (8 spheres), boxes.txt (spheres + boxes), 4K fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import parse_scene_text, separate_objects, _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
W, H = 3840, 2160
pool = open(os.path.join(ROOT, "scenes", "pool.txt")).read()
scenes = {"pool + 2 spheres (8 sph, 1 pln)": pool + "sph -3.0 0.0 -4.0 1.0 1\nsph 3.0 0.0 -4.0 1.0 2\n",
          "pool - 2 spheres (4 sph, 1 pln)": "\n".join(l for l in pool.splitlines() if not l.startswith("sph 0.0 0.0 -4.0") and not l.startswith("sph 2.0 0.0 0.0")) + "\n",
          "boxes.txt": open(os.path.join(ROOT, "scenes", "boxes.txt")).read()}
for name, text in scenes.items():
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    cam.setup(W / H)
    with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
        best = 1e9
        for _ in range(4):
            gs.render(cam, W, H, 5, 3, precision="fp32", jitter=1234, flags=_lib.RT_FLAG_TIMING)
            best = min(best, gs.last_stats["shade_ms"])
        print(f"{name}: shade {best:.3f} ms (RT_SHADE={os.environ.get('RT_SHADE', 'ls')})")
