#!/usr/bin/env python3
"""Compute side of the strong-scaling curve with k frames in flight: ONE rank's row set (part 0 of N interleaved
8-row blocks) of a workload's frame rendered on one GPU, frames alternating between k scene handles (own streams and
workspaces).  Prints ms per frame and the compute-side efficiency against the full frame with 2 in flight / N.
    python tools/part_in_flight_probe.py [workload] """
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import load_workload
from raytracer_amd import _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
wl = sys.argv[1] if len(sys.argv) > 1 else "pool4k"
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(wl)
n_sh, depth = st.root_number_shadow_rays, st.max_recursions
cam.setup(W / H)
hip = ctypes.CDLL("libamdhip64.so")
esz = 12 if precision == "fp32" else 24
packed = pack_scene(mats, surfs, lights, st.background_color)
KMAX = 4
hs = [GpuScene(packed) for _ in range(KMAX)]
bufs = []
for _ in range(KMAX):
    b = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(b), W * H * esz) == 0; bufs.append(b)
frames = int(os.environ.get("FRAMES", 48 if wl != "synth4k" else 12))
base = None
for n in (1, 2, 4, 8):
    for k in (1, 2, 3, 4):
        def go(f):
            for i in range(f):
                hs[i % k].render(cam, W, H, n_sh, depth, precision=precision, jitter=1234, block_rows=8, part=0, n_parts=n,
                                 out_ptr=bufs[i % k].value, flags=_lib.RT_FLAG_ASYNC)
            for h in hs[:k]:
                h.sync()
        go(2 * k)
        t0 = time.perf_counter(); go(frames); ms = (time.perf_counter() - t0) / frames * 1e3
        if n == 1 and k == 2:
            base = ms
        eff = f"  compute-side efficiency vs (N=1, 2 in flight)/N: {base / n / ms:.2f}" if base else ""
        print(f"{wl} N={n} part 0, {k} in flight: {ms:.4f} ms/frame{eff}", flush=True)
for h in hs:
    h.close()
