#!/usr/bin/env python3
"""Kernel-time probe without torch: renders a workload a few times through the C-ABI with
RT_FLAG_TIMING and prints per-kernel-class milliseconds (HIP events on the library's stream).

    python tools/quick_bench.py [workload] [--frames N] [--check]
Frames go into a device buffer (hipMalloc through ctypes) back to back, as in bench.py's timed region: with a host copy
of the image after every frame the GPU idles for tens of milliseconds between frames, its clocks drop and the same
kernels measure ~9 % slower (--host-out keeps that behaviour).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_BANDS", "1")      # kernel-time probe: one frame = one set of launches (no row bands for the host copy)

import numpy as np  # noqa: E402

from bench import WORKLOADS, load_workload, PEAK_TFLOPS  # noqa: E402
from raytracer_amd import _lib  # noqa: E402
from raytracer_amd.flops import shade_kernel_flops, rays_traced_by_shade_kernel  # noqa: E402
from raytracer_amd.packing import pack_scene  # noqa: E402
from raytracer_amd.render import GpuScene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="pool4k", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--scale", type=int, default=1, help="divide width/height by this")
    ap.add_argument("--check", action="store_true", help="compare a 1/4-size frame with the oracle")
    ap.add_argument("--n-shadow", type=int, default=None, help="override the shadow-ray root (cost breakdowns)")
    ap.add_argument("--depth", type=int, default=None, help="override the recursion cap")
    ap.add_argument("--brute", action="store_true", help="RT_FLAG_BRUTE_FORCE frames: every surface tested for every shadow sample (the flop model's work; "
                                                         "the run profiles/r03_*_brute_kernel_stats.csv comes from)")
    ap.add_argument("--host-out", action="store_true", help="copy every frame to the host (GPU idles between frames)")
    args = ap.parse_args()
    cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(args.workload)
    W //= args.scale; H //= args.scale
    if args.n_shadow is not None:
        st.root_number_shadow_rays = args.n_shadow
    if args.depth is not None:
        st.max_recursions = args.depth
    n_shadow, max_depth = st.root_number_shadow_rays, st.max_recursions
    cam.setup(W / H)
    packed = pack_scene(mats, surfs, lights, st.background_color)
    n_sph, n_pln, n_box = packed.counts
    dev_ptr = None
    if not args.host_out:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        p = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(W * H * 3 * (4 if precision == "fp32" else 8)))
        if rc != 0 or not p.value:
            raise SystemExit(f"hipMalloc failed ({rc})")
        dev_ptr = p.value
    with GpuScene(packed) as gs:
        best = None
        flags = _lib.RT_FLAG_TIMING | (_lib.RT_FLAG_BRUTE_FORCE if args.brute else 0)
        for f in range(args.frames):
            t0 = time.perf_counter()
            img = gs.render(cam, W, H, n_shadow, max_depth, precision=precision, jitter=1234, flags=flags, out_ptr=dev_ptr)
            wall = (time.perf_counter() - t0) * 1e3
            s = gs.last_stats
            if best is None or s["total_ms"] < best["total_ms"]:
                best = dict(s, wall_ms=wall)
        if dev_ptr is not None:
            img = gs.render(cam, W, H, n_shadow, max_depth, precision=precision, jitter=1234, flags=flags)
        fr_, fc_ = rays_traced_by_shade_kernel(best)
        flops = shade_kernel_flops(n_sph=n_sph, n_pln=n_pln, n_box=n_box, n_lights=len(lights), n_shadow=n_shadow,
                                   hits=best["hits"], segments=max(best["segments"], 0), rays=fr_, children=fc_)
        rays = best["rays"] + best["shadow_rays"]
        out = {"workload": (desc if args.scale == 1 else f"{desc} /{args.scale}") + (" [RT_FLAG_BRUTE_FORCE]" if args.brute else ""), "frame_ms": round(best["total_ms"], 3),
               "shade_ms": round(best["shade_ms"], 3), "trace_ms": round(best["trace_ms"], 3), "other_ms": round(best["other_ms"], 3),
               "Mrays_per_s": round(rays / best["total_ms"] / 1e3, 1),
               "shade_TFLOPs": round(flops / best["shade_ms"] / 1e9, 2),
               "shade_frac_of_peak": round(flops / best["shade_ms"] / 1e9 / PEAK_TFLOPS[precision], 4),
               "rays": best["rays"], "hits": best["hits"], "shadow_rays": best["shadow_rays"], "level_rays": best["level_rays"],
               "level_hits": best["level_hits"], "img_mean": float(np.mean(img))}
        print(json.dumps(out))
        if args.check:
            from oracle import oracle
            from raytracer_amd.image import quantize_u8
            oracle.set_threads(min(16, os.cpu_count() or 1))
            w, h = W // 4, H // 4
            cam.setup(w / h)
            ref = oracle.render(cam, st, mats, surfs, lights, w, h, jitter=("pixel", 1234))
            img = gs.render(cam, w, h, n_shadow, max_depth, precision=precision, jitter=1234)
            d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
            rel = np.abs(img - ref) / np.maximum(np.abs(ref), 1e-12)
            print(json.dumps({"check": f"{w}x{h}", "max_8bit_diff": int(d8.max()), "channels_gt_1": int((d8 > 1).sum()),
                              "channels": int(d8.size), "max_rel": float(rel.max()), "mean_abs": float(np.abs(img - ref).mean())}))


if __name__ == "__main__":
    main()
