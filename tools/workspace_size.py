#!/usr/bin/env python3
"""Device memory a scene's workspace takes for one frame size (hipMemGetInfo before / after the first frames):
   tools/workspace_size.py [workload]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, load_workload  # noqa: E402
from raytracer_amd.packing import pack_scene  # noqa: E402
from raytracer_amd.render import GpuScene  # noqa: E402
import torch  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def free_bytes():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


name = sys.argv[1] if len(sys.argv) > 1 else "pool8k"
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(name)
cam.setup(W / H)
n, depth = st.root_number_shadow_rays, st.max_recursions
with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
    out = torch.empty((H, W, 3), dtype=torch.float32 if precision == "fp32" else torch.float64, device="cuda")
    torch.cuda.synchronize()
    before = free_bytes()
    for _ in range(2):
        gs.render(cam, W, H, n, depth, precision=precision, jitter=1, out_ptr=out.data_ptr())
    after = free_bytes()
    lr = gs.last_stats["level_rays"][:depth + 1]
    print(json.dumps({"workload": desc, "pixels": W * H, "workspace_GB": round((before - after) / 1e9, 3),
                      "largest_bounce_level_rays_per_pixel": round(max(lr[1:] or [0]) / (W * H), 3)}))
