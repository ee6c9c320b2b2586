#!/usr/bin/env python3
"""Per-frame shade/trace kernel times of back-to-back frames (sustained-clock behaviour): blocking
frames with a host copy in between vs asynchronous frames into device memory."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from bench import load_workload
from raytracer_amd import _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload("pool4k")
cam.setup(W / H)
with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
    t = []
    for f in range(12):
        gs.render(cam, W, H, 5, 3, precision="fp32", jitter=1234, flags=_lib.RT_FLAG_TIMING)
        t.append(round(gs.last_stats["shade_ms"], 3))
    print("blocking + D2H copy:", t)
    import ctypes
    lib = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.c_void_p()
    assert lib.hipMalloc(ctypes.byref(buf), W * H * 12) == 0
    stats = [_lib.Stats() for _ in range(30)]
    for s in stats:
        gs.render(cam, W, H, 5, 3, precision="fp32", jitter=1234, out_ptr=buf.value, flags=_lib.RT_FLAG_TIMING | _lib.RT_FLAG_ASYNC, stats=s)
    gs.sync()
    print("async back-to-back:", [round(s.shade_ms, 3) for s in stats])
