#!/usr/bin/env python3
"""One non-retried frame with small queues: prints the overflow message (what a level asked for) or 'fits'.
   RT_QUEUE_SLOTS_PER_PIXEL=0.5 tools/overflow_probe.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_workload  # noqa: E402
from raytracer_amd import _lib  # noqa: E402
from raytracer_amd.packing import pack_scene  # noqa: E402
from raytracer_amd.render import GpuScene  # noqa: E402
import torch  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "pool4k"
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(name)
cam.setup(W / H)
with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
    out = torch.empty((H, W, 3), dtype=torch.float32 if precision == "fp32" else torch.float64, device="cuda")
    try:
        gs.render(cam, W, H, st.root_number_shadow_rays, st.max_recursions, precision=precision, jitter=1, out_ptr=out.data_ptr(), flags=_lib.RT_FLAG_ASYNC)
        gs.sync()
        print("fits")
    except _lib.RtError as e:
        print(e)
