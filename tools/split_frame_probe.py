#!/usr/bin/env python3
"""One frame rendered by n scene handles on ONE GPU at once (rt_render_multi, interleaved row blocks): does splitting a
frame over streams shorten it the way two frames in flight raise the throughput?  tools/split_frame_probe.py [workload]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import load_workload
from raytracer_amd import _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene, render_multi
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(sys.argv[1] if len(sys.argv) > 1 else "pool4k")
n_sh, depth = st.root_number_shadow_rays, st.max_recursions
cam.setup(W / H)
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(buf), W * H * 3 * (4 if precision == "fp32" else 8)) == 0
packed = pack_scene(mats, surfs, lights, st.background_color)
for n in (1, 2, 3, 4):
    scenes = [GpuScene(packed) for _ in range(n)]
    for _ in range(6):
        render_multi(scenes, cam, W, H, n_sh, depth, precision=precision, jitter=1234, out_ptr=buf.value)
    t0 = time.perf_counter()
    for _ in range(30):
        render_multi(scenes, cam, W, H, n_sh, depth, precision=precision, jitter=1234, out_ptr=buf.value)
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"{n} handle(s): {ms:.3f} ms per frame (blocking calls, wall clock)")
    for s in scenes:
        s.close()
