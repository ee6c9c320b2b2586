import ctypes, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import load_workload
from raytracer_amd import _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload("pool4k")
n_sh, depth = st.root_number_shadow_rays, st.max_recursions
cam.setup(W / H)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
packed = pack_scene(mats, surfs, lights, st.background_color)
hip = ctypes.CDLL("libamdhip64.so")
def raw_stream():
    s = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0; return s.value   # hipStreamNonBlocking
tiles = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
def run(kind):
    a, b = GpuScene(packed), GpuScene(packed)
    keep = []
    if kind == "torch":
        ss = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        a.set_stream(ss[0].cuda_stream); b.set_stream(ss[1].cuda_stream); keep = ss
    elif kind == "torch_hi":
        ss = [torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)]
        a.set_stream(ss[0].cuda_stream); b.set_stream(ss[1].cuda_stream); keep = ss
    elif kind == "raw":
        r = [raw_stream(), raw_stream()]
        ss = [torch.cuda.ExternalStream(r[0]), torch.cuda.ExternalStream(r[1])]
        a.set_stream(r[0]); b.set_stream(r[1]); keep = ss
    def go(n):
        for k in range(n):
            (a if k % 2 == 0 else b).render(cam, W, H, n_sh, depth, precision=precision, jitter=1234, out_ptr=tiles[k % 2].data_ptr(), flags=_lib.RT_FLAG_ASYNC)
        a.sync(); b.sync(); torch.cuda.synchronize()
    go(8)
    t0 = time.perf_counter(); go(40); ms = (time.perf_counter() - t0) / 40 * 1e3
    print(kind, round(ms, 4), [hex(getattr(x, "cuda_stream", 0)) for x in keep])
    a.close(); b.close()
for kind in ("private", "torch", "torch_hi", "raw", "private"):
    run(kind)
