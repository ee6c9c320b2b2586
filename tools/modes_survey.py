#!/usr/bin/env python3
"""Every way into the hot path, on randomised scenes (small / authored / 7-16 spheres / large): numpy-stream jitter and
u8 output against the oracle, row tiles, RT_FLAG_BRUTE_FORCE, rt_render_static and rt_render_multi against the plain
frame, rt_nearest_hit / rt_shadow_transmission on random rays against the oracle (bit-exact).
tools/modes_survey.py [n_seeds [first_seed]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from test_gpu_random_scenes import random_scene_text
from test_gpu_many_spheres import many_spheres_text
from authored_survey import authored_scene_text
from raytracer_amd import parse_scene_text, separate_objects, _lib
from raytracer_amd.render import GpuScene, render_multi
from raytracer_amd.packing import rows_in_part
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
W, H = 72, 56


def q(img):
    return quantize_u8(img).astype(int)


for seed in range(FIRST, FIRST + N):
    kind = seed % 4
    if kind == 0: text = random_scene_text(11000 + seed, seed % 8 == 0)
    elif kind == 1: text = authored_scene_text(seed)
    elif kind == 2: text = many_spheres_text(7 + seed % 10, 12000 + seed, seed % 8 == 2)
    else: text = random_scene_text(60000 + seed, seed % 8 == 3, big=True)
    cam, st, objs = parse_scene_text(text)
    mats, surfs, lights = separate_objects(objs)
    n, d = int(st.root_number_shadow_rays), int(st.max_recursions)
    errs = []
    ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
    with GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs, GpuScene.from_objects(mats, surfs, lights, st.background_color) as gs2:
        kw = dict(jitter=("pixel", seed))
        f64 = gs.render(cam, W, H, n, d, precision="fp64", **kw).copy(); s64 = dict(gs.last_stats)
        f32 = gs.render(cam, W, H, n, d, precision="fp32", **kw).copy(); s32 = dict(gs.last_stats)
        if (np.abs(f64 - ref) / np.maximum(np.abs(ref), 1e-3)).max() > 1e-9: errs.append("fp64 vs oracle")
        if np.abs(q(f32) - q(ref)).max() > 1: errs.append("fp32 vs oracle")
        # numpy-stream jitter (fp64, small frames: the oracle draws the same stream)
        refA = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("numpy", seed))
        fA = gs.render(cam, W, H, n, d, precision="fp64", jitter=("numpy", seed))
        if (np.abs(fA - refA) / np.maximum(np.abs(refA), 1e-3)).max() > 1e-9: errs.append("numpy-stream fp64 vs oracle")
        # u8 output of both precisions against the quantised oracle image
        for prec in ("fp64", "fp32"):
            u8 = gs.render(cam, W, H, n, d, precision=prec, out_format="u8", **kw)
            if np.abs(u8.astype(int) - q(ref)).max() > (0 if prec == "fp64" else 1):
                # fp64: a value within rounding of a quantisation step may truncate differently
                dd = np.abs(u8.astype(int) - q(ref))
                if dd.max() > 1 or (prec == "fp64" and (dd > 0).sum() > 3): errs.append(f"u8 {prec}: max {dd.max()}, {(dd > 0).sum()} differ")
        # row tiles (3 parts, blocks of 8 rows) reassemble the frame
        for prec, full, tol in (("fp64", f64, 1e-12), ("fp32", f32, 2e-6)):
            asm = np.zeros_like(full)
            for part in range(3):
                rows = rows_in_part(H, 8, part, 3)
                asm[rows] = gs.render(cam, W, H, n, d, precision=prec, block_rows=8, part=part, n_parts=3, **kw)
            if np.abs(asm - full).max() > tol * max(1.0, np.abs(full).max()): errs.append(f"tiles {prec}: {np.abs(asm - full).max():.2e}")
        # brute-force flag: same image (fp32 within a level: other pairs reach the re-test), same counters
        fb = gs.render(cam, W, H, n, d, precision="fp32", flags=_lib.RT_FLAG_BRUTE_FORCE, **kw); sb = dict(gs.last_stats)
        if np.abs(q(fb) - q(ref)).max() > 1: errs.append("brute-force fp32 vs oracle")
        if any(sb[k] != s32[k] for k in ("rays", "hits", "shadow_rays")): errs.append("brute-force counters")
        if any(s64[k] != s32[k] for k in ("rays", "hits", "shadow_rays")): errs.append("fp32 counters")
        # the entry point that allocates nothing
        t = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        try:
            gs.render(cam, W, H, n, d, precision="fp32", out_ptr=t.data_ptr(), static=True, **kw)
            torch.cuda.synchronize()
            if np.abs(t.cpu().numpy() - f32).max() > 2e-6 * max(1.0, np.abs(f32).max()): errs.append(f"static: {np.abs(t.cpu().numpy() - f32).max():.2e}")
        except _lib.RtError as e:          # its fixed queues are a documented limit; anything else is a failure
            if "static queue" not in str(e): errs.append(f"static: {e}")
        # two handles, one frame
        fm, _ = render_multi([gs, gs2], cam, W, H, n, d, precision="fp32", **kw)
        if np.abs(fm - f32).max() > 2e-6 * max(1.0, np.abs(f32).max()): errs.append(f"multi: {np.abs(fm - f32).max():.2e}")
        # helpers on random rays
        rng = np.random.default_rng(seed)
        M = 4000
        o = rng.uniform(-6, 6, (M, 3)); o[:, 1] = rng.uniform(-1, 8, M)
        dr = rng.normal(size=(M, 3)); dr /= np.linalg.norm(dr, axis=1)[:, None]
        t1, i1, n1 = gs.nearest_hit(o, dr); t2, i2, n2 = oracle.nearest_hit(surfs, o, dr)
        if not (np.array_equal(t1, t2) and np.array_equal(i1, i2) and np.array_equal(n1, n2)): errs.append("nearest_hit not bit-exact")
        md = rng.uniform(0.5, 15, M)
        T1 = gs.shadow_transmission(o, dr, md); T2 = oracle.shadow_transmission(mats, surfs, o, dr, md)
        if not np.array_equal(T1, T2): errs.append(f"shadow_transmission: {np.abs(T1 - T2).max():.2e}")
    if errs:
        bad += 1
        print("seed", seed, "kind", kind, "surfaces", len(surfs), "n", n, "depth", d, ":", "; ".join(errs), flush=True)
print(f"{N} scenes, {bad} with a failure")
