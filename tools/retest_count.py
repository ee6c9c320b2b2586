#!/usr/bin/env python3
"""(hit, light) pairs a workload's fp32 frame sends to the exact re-test, and the frame's kernel times:
tools/retest_count.py [workload] [--scale k]   (RT_NEAR_PAIRS=0 etc. for A/B)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("RT_BANDS", "1")
from bench import load_workload
from raytracer_amd import _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
wl = sys.argv[1] if len(sys.argv) > 1 else "synth4k"
scale = int(sys.argv[sys.argv.index("--scale") + 1]) if "--scale" in sys.argv else 1
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(wl)
W //= scale; H //= scale
cam.setup(W / H)
with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
    for _ in range(3):
        gs.render(cam, W, H, st.root_number_shadow_rays, st.max_recursions, precision="fp32", jitter=1234, flags=_lib.RT_FLAG_TIMING)
    s = gs.last_stats
    print(f"{desc}: hits {s['hits']}, (hit, light) pairs {s['hits'] * len(lights)}, re-tested {s['retested']} ({100.0 * s['retested'] / (s['hits'] * len(lights)):.3f} %), "
          f"frame {s['total_ms']:.3f} ms, shade {s['shade_ms']:.3f}, trace {s['trace_ms']:.3f}")
