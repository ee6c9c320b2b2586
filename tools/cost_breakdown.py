#!/usr/bin/env python3
"""Shade-kernel cost breakdown on the pool scene: per-hit vs per-(hit,light) vs per-sample cost,
by rendering variants with 0/1/5 lights and 1x1 / 5x5 shadow grids (HIP-event kernel times)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import parse_scene_text, separate_objects, _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene

text = open(os.path.join(ROOT, "scenes", "pool.txt")).read()
W, H = 3840, 2160
for n_lights in (0, 1, 5):
    lines, seen = [], 0
    for ln in text.splitlines():
        if ln.startswith("lgt"):
            seen += 1
            if seen > n_lights:
                continue
        lines.append(ln)
    cam, st, objs = parse_scene_text("\n".join(lines))
    mats, surfs, lights = separate_objects(objs)
    cam.setup(W / H)
    with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
        for n, depth in ((1, 3), (5, 3), (1, 0)):
            best = None
            for _ in range(3):
                gs.render(cam, W, H, n, depth, precision="fp32", jitter=1234, flags=_lib.RT_FLAG_TIMING)
                s = gs.last_stats
                if best is None or s["shade_ms"] < best["shade_ms"]:
                    best = s
            print(json.dumps({"lights": n_lights, "n_shadow": n, "depth": depth, "shade_ms": round(best["shade_ms"], 3),
                              "trace_ms": round(best["trace_ms"], 3), "hits": best["hits"]}))
