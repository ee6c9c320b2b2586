import sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import load_workload
from raytracer_amd import _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
wl = sys.argv[1] if len(sys.argv) > 1 else "pool4k"
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(wl)
n_sh, depth = st.root_number_shadow_rays, st.max_recursions
cam.setup(W / H)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
packed = pack_scene(mats, surfs, lights, st.background_color)
dt = torch.float32 if precision == "fp32" else torch.float64
for n in (1, 2, 3, 4, 2, 3):
    hs = [GpuScene(packed) for _ in range(n)]
    tiles = [torch.empty((H, W, 3), dtype=dt, device=dev) for _ in range(n)]
    def go(k_):
        for k in range(k_):
            hs[k % n].render(cam, W, H, n_sh, depth, precision=precision, jitter=1234, out_ptr=tiles[k % n].data_ptr(), flags=_lib.RT_FLAG_ASYNC)
        for h in hs: h.sync()
        torch.cuda.synchronize()
    go(12)
    t0 = time.perf_counter(); go(60); ms = (time.perf_counter() - t0) / 60 * 1e3
    print(wl, n, "in flight:", round(ms, 4))
    for h in hs: h.close()
