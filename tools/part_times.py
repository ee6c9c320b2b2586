#!/usr/bin/env python3
"""Render time of ONE rank's row set of the pool 4K frame for N = 1, 2, 4, 8 ranks (asynchronous
back-to-back frames into device memory, wall clock around 30 frames): the compute side of the
strong-scaling curve, without the gather."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import load_workload
from raytracer_amd import _lib
from raytracer_amd.packing import pack_scene
from raytracer_amd.render import GpuScene
cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(sys.argv[1] if len(sys.argv) > 1 else "pool4k")
n_sh, depth = st.root_number_shadow_rays, st.max_recursions
cam.setup(W / H)
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(buf), W * H * 12) == 0
with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as gs:
    base = None
    for n in (1, 2, 4, 8):
        br = int(os.environ.get("BR", 0)) or [b for b in range(16, 3, -1) if H % b == 0 and (H // b) % n == 0][0]
        for part in sorted({0, n - 1}):
            for _ in range(3):
                gs.render(cam, W, H, n_sh, depth, precision=precision, jitter=1234, block_rows=br, part=part, n_parts=n,
                          out_ptr=buf.value, flags=_lib.RT_FLAG_ASYNC)
            gs.sync()
            t0 = time.perf_counter()
            for _ in range(30):
                gs.render(cam, W, H, n_sh, depth, precision=precision, jitter=1234, block_rows=br, part=part, n_parts=n,
                          out_ptr=buf.value, flags=_lib.RT_FLAG_ASYNC)
            gs.sync()
            ms = (time.perf_counter() - t0) / 30 * 1e3
            base = base or ms
            print(f"N={n} part {part} block_rows {br}: {ms:.3f} ms/frame  (ideal {base / n:.3f}, compute-side efficiency {base / n / ms:.2f})")
    # kernel-time sums vs first-start -> last-end of one frame (gaps between launches), N = 8 row set
    for n in (1, 8):
        br = [b for b in range(16, 3, -1) if H % b == 0 and (H // b) % n == 0][0]
        s = _lib.Stats()
        for _ in range(3):
            gs.render(cam, W, H, n_sh, depth, precision=precision, jitter=1234, block_rows=br, part=0, n_parts=n,
                      out_ptr=buf.value, flags=_lib.RT_FLAG_ASYNC | _lib.RT_FLAG_TIMING, stats=s)
            gs.sync()
        print(f"N={n}: trace {s.trace_ms:.3f} ms in {s.trace_launches} launches, shade {s.shade_ms:.3f} ms in {s.shade_launches}, "
              f"other {s.other_ms:.3f}, first start -> last end {s.total_ms:.3f} ms; level hits {list(s.level_hits[:depth + 1])}")
# two scene handles (own streams / workspaces), frames alternating between them: tails of one frame's
# small launches are filled by the other frame's kernels
buf2 = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(buf2), W * H * 12) == 0
with GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as a, GpuScene(pack_scene(mats, surfs, lights, st.background_color)) as b:
    for n in (1, 2, 4, 8):
        br = int(os.environ.get("BR", 0)) or [x for x in range(16, 3, -1) if H % x == 0 and (H // x) % n == 0][0]
        def go(frames):
            for k in range(frames):
                (a if k % 2 == 0 else b).render(cam, W, H, n_sh, depth, precision=precision, jitter=1234, block_rows=br, part=0, n_parts=n,
                                                out_ptr=(buf if k % 2 == 0 else buf2).value, flags=_lib.RT_FLAG_ASYNC)
            a.sync(); b.sync()
        go(4)
        t0 = time.perf_counter(); go(40); ms = (time.perf_counter() - t0) / 40 * 1e3
        print(f"two frames in flight, N={n} part 0: {ms:.3f} ms/frame")
