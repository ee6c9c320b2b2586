#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: frames of BASELINE.json's headline workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one frame: every rank renders its interleaved row blocks of the frame into device memory
(the scene is already resident in HBM) and, for N > 1, the tiles are gathered to rank 0 over RCCL.
The frame is fixed as N grows (strong scaling).  Rank 0 prints ONE JSON line.
For every N several frames (RT_BENCH_IN_FLIGHT, default 3) are in flight on each rank (one scene handle and one
non-blocking stream per frame in flight: the next frames' level 0 fills the tails of this frame's small launches; for
N > 1 the gather of frame k also overlaps the render of k + 1 ...), so `value` / `ms_per_step` are throughput numbers; `one_frame_at_a_time` (N = 1) carries the frame time of the
same frames rendered one after the other on one stream, and RT_BENCH_PIPELINE=0 makes that the headline.

metric/value: Mrays/s, rays = radiance rays + shadow rays of the frame, whole job.
roofline: the dominant kernel (k_shade_ls / k_shade_fast / k_shade_exact) — algorithmic flops of its launches
  (raytracer_amd/flops.py) / their summed duration, measured with HIP events on the
  stream the kernels run on (rt_stats, RT_FLAG_TIMING) on every --timing-every-th frame (N = 1: of the serial loop, so that
  overlap does not distort them).  The bound is the
  vector ALU (no MFMA-shaped work, HBM traffic is one framebuffer write), peak = 157.3 TFLOP/s fp32
  / 78.6 fp64 (MI355X_MICROARCH.md chip table).
cpu_baseline: the C oracle (oracle/rt_oracle.c, a port of the reference's algorithm) timed on this
  box's host cores on a bounded sample of the same workload; rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene file, W, H, n_shadow, max_depth, precision, description)
    "pool4k": ("pool.txt", 3840, 2160, 5, 3, "fp32", "pool.txt 3840x2160, 5x5 shadow rays, depth 3, fp32 (BASELINE configs[2])"),
    "original1080": ("original.txt", 1920, 1080, None, None, "fp64", "original.txt 1920x1080, file settings (3x3, depth 5), fp64 (BASELINE configs[1])"),
    "pool8k": ("pool.txt", 7680, 4320, 8, 6, "fp32", "pool.txt 7680x4320, 8x8 shadow rays, depth 6, fp32 (BASELINE configs[4])"),
    "synth4k": ("synth:4096:512", 3840, 2160, 3, 3, "fp32", "synthetic 4096 spheres + 512 boxes 3840x2160, 3x3 shadow rays, depth 3, fp32 (BASELINE configs[3])"),
    "pool500": ("pool.txt", 500, 500, None, None, "fp64", "pool.txt 500x500 file settings, fp64 (BASELINE configs[0] shape)"),
}
PEAK_TFLOPS = {"fp32": 157.3, "fp64": 78.6}


def load_workload(name):
    from raytracer_amd import load_scene, parse_scene_text, separate_objects
    from raytracer_amd.synth import synthetic_scene_text
    scene, W, H, n, depth, precision, desc = WORKLOADS[name]
    if scene.startswith("synth:"):
        _, ns, nb = scene.split(":")
        cam, st, objs = parse_scene_text(synthetic_scene_text(int(ns), int(nb), shadow_rays=n, max_recursions=depth))
        mats, surfs, lights = separate_objects(objs)
    else:
        cam, st, mats, surfs, lights = load_scene(os.path.join(ROOT, "scenes", scene), n, depth)
    return cam, st, mats, surfs, lights, W, H, precision, desc


_raw_streams = []


def new_stream(dev):
    """A hipStreamNonBlocking stream of our own, wrapped for torch.  Two torch.cuda.Stream() objects of the default
    priority did NOT run their kernels concurrently on this stack (two frames in flight measured 1.65 ms per frame on
    them, 1.52 on raw non-blocking streams or on the library's private ones: tools/streams_probe.py), so the streams the
    pipelined N > 1 path renders and gathers on are created through the runtime directly."""
    import ctypes
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1)      # 1 = hipStreamNonBlocking
    if rc != 0 or not raw.value:
        return torch.cuda.Stream(dev)
    _raw_streams.append(raw)
    return torch.cuda.ExternalStream(raw.value, device=dev)


def cpu_baseline(name, gs, cam, st, mats, surfs, lights, W, H, precision):
    """Time the oracle on a bounded sample (a half-resolution frame of the same workload: ~30 s of CPU
    work spread over the threads) and check the GPU render of the same sample against it."""
    import numpy as np
    from oracle import oracle
    from raytracer_amd.image import quantize_u8
    threads = min(16, os.cpu_count() or 1)
    oracle.build()
    oracle.set_threads(threads)
    div = 2 if name != "synth4k" else 40
    w, h = max(W // div, 16), max(H // div, 9)
    cam.setup(w / h)
    t0 = time.perf_counter()
    ref, ost = oracle.render(cam, st, mats, surfs, lights, w, h, jitter=("pixel", 1234), return_stats=True)
    dt = time.perf_counter() - t0
    rays = ost["rays"] + ost["shadow_rays"]
    img = gs.render(cam, w, h, st.root_number_shadow_rays, st.max_recursions, precision=precision, jitter=1234)
    cam.setup(W / H)
    d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
    rel = float((np.abs(img - ref) / np.maximum(np.abs(ref), 1e-12)).max())
    base = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{WORKLOADS[name][0]} {w}x{h} (1/{div * div} of the frame's pixels), same shadow rays/depth, "
                      f"1 frame, {rays} rays in {dt:.2f} s, C oracle with OpenMP"}
    parity = {"sample": f"{w}x{h}", "max_8bit_diff": int(d8.max()), "channels_gt_1_level": int((d8 > 1).sum()),
              "channels": int(d8.size), "max_rel_err": rel}
    # the reference's own figure, measured in the build container (the reference cannot travel)
    base["reference_numpy"] = {"value": 0.763, "unit": "Mrays/s", "cores": 1,
                               "sample": "pool.txt 500x500 file settings, render_vectorized of the reference itself, 8 vCPU Xeon container"}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pool4k", choices=sorted(WORKLOADS))
    ap.add_argument("--block-rows", type=int, default=0,
                    help="rows per interleaved block (0 = 8, the height of the kernels' pixel tiles)")
    ap.add_argument("--out-format", default="native", choices=["native", "u8"],
                    help="frame format in device memory, every N: native = float32 / float64 linear radiance (what render_vectorized returns; "
                         "the default), u8 = the 8-bit image of save_image quantised on the device (SURVEY 8 f1: the gather moves 4x fewer bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timing", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="bracket the kernels of every k-th timed frame with HIP events (the events cost ~3 %% of a 4K frame)")
    ap.add_argument("--prewarm", type=int, default=16, help="untimed setup frames before the warmup steps (workspace growth, clock ramp)")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the comparison of the gathered frame with a full-frame render on rank 0 (outside the timed region)")
    ap.add_argument("--brute-frames", type=int, default=3, help="fp32, N = 1: frames rendered with RT_FLAG_BRUTE_FORCE after the timed region for roofline.frac (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: run the ranks as a child torchrun job (a child
        # process, not an exec — nothing in this process has touched the GPU yet)
        import socket
        import subprocess
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracer_amd import _lib
    from raytracer_amd.dist import RowTiling, cache_row_indices, gather_tiles, new_scratch
    from raytracer_amd.flops import shade_kernel_flops, flops_per_query, rays_traced_by_shade_kernel
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.render import GpuScene

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # Rehearsal hooks for a one-GPU box: RT_BENCH_BACKEND=gloo RT_BENCH_ONE_DEVICE=1 run all ranks on
    # cuda:0 and gather through host memory.  The real multi-GPU run uses nccl (= RCCL over xGMI).
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if os.environ.get("RT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cam, st, mats, surfs, lights, W, H, precision, desc = load_workload(args.workload)
    if args.block_rows <= 0:
        args.block_rows = 8       # = the height of the kernels' 8x8 pixel tiles; rank loads differ by <= 8 rows
    n_shadow, max_depth = int(st.root_number_shadow_rays), int(st.max_recursions)
    cam.setup(W / H)
    packed = pack_scene(mats, surfs, lights, st.background_color)
    gs = GpuScene(packed, device=local_rank)
    # The library renders on a stream of its own unless it is given one (rt_scene_set_stream; NULL = its private
    # non-blocking stream, which torch's default stream does NOT order against).  N = 1 needs no sharing: the timed
    # region ends with a device-wide synchronize.  N > 1: the render and the gather run on ONE explicit torch stream
    # (or on two with events between them when pipelined), so the gather can never read a half-rendered tile.
    # Frames in flight (every N): frames alternate between `depth` scene handles (each with its own workspace, render
    # stream and tile buffer), so the kernels of the next frames fill the tails of frame k's small launches, and for
    # N > 1 the gather of frame k (comm stream) overlaps the render of frame k+1.  Measured on one GPU with one rank's
    # row set (tools/part_in_flight_probe.py): 3 in flight beats 2 by 2 % at N = 1 and by 6-9 % on the row sets of
    # N = 4 / 8 (whose launches are short and latency bound); 4 is worse than 3 (the runtime has 4 hardware queues).
    # RT_BENCH_PIPELINE=0 serialises everything on one stream (frame latency).
    pipelined = os.environ.get("RT_BENCH_PIPELINE", "1") != "0"
    depth = max(2, min(8, int(os.environ.get("RT_BENCH_IN_FLIGHT", "3")))) if pipelined else 1
    serial_stream = new_stream(dev) if (world > 1 and not pipelined) else None
    if serial_stream is not None:
        gs.set_stream(serial_stream.cuda_stream)

    tiling = cache_row_indices(RowTiling(H, W, world, args.block_rows), dev)
    u8 = args.out_format == "u8"
    dtype = torch.uint8 if u8 else (torch.float32 if precision == "fp32" else torch.float64)
    common = dict(precision=precision, jitter=1234, out_format="u8" if u8 else None)      # of every render call into a tile / frame buffer
    n_buf = depth
    tiles = [torch.empty((tiling.max_rows, W, 3), dtype=dtype, device=dev) for _ in range(n_buf)]
    tile = tiles[0]
    frame = torch.empty((H, W, 3), dtype=dtype, device=dev) if (world > 1 and rank == 0) else None
    scratch = new_scratch(tile, world) if (world > 1 and rank == 0) else None
    every = max(1, args.timing_every)

    def flags_of(k):
        if world == 1 and pipelined:      # (N = 1: the event-timed frames come from a serial loop after the timed region)
            return _lib.RT_FLAG_ASYNC
        return _lib.RT_FLAG_ASYNC | (_lib.RT_FLAG_TIMING if (not args.no_timing and k % every == 0) else 0)
    scenes = [gs] + [GpuScene(packed, device=local_rank) for _ in range(depth - 1)]      # (N = 1: each handle renders on its own private non-blocking stream)
    if pipelined and world > 1:
        render_streams = [new_stream(dev) for _ in range(depth)]
        comm_stream = new_stream(dev)
        for sc_, st_ in zip(scenes, render_streams):
            sc_.set_stream(st_.cuda_stream)
        rendered = [torch.cuda.Event() for _ in range(depth)]
        tile_free = [None] * depth
    frame_no = [0]
    gather_events = []      # (start, end) on the stream the gather runs on, for the frames whose kernels are event-timed too

    def timed_gather(stream, k, *a, **kw):
        """gather_tiles on `stream` (already current); every `every`-th frame is bracketed with events so that the N > 1
        line can say how long the gather + reassembly took on this rank (on rank 0 that includes waiting for the
        slowest peer's tile: compare with compute_ms to see which side bounds the frame)."""
        if args.no_timing or k % every != 0:
            return gather_tiles(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r = gather_tiles(*a, **kw)
        e1.record(stream)
        gather_events.append((e0, e1))
        return r

    def step(stats):
        k = frame_no[0]; frame_no[0] += 1
        flags = flags_of(k)
        if not pipelined:
            gs.render(cam, W, H, n_shadow, max_depth, **common, block_rows=args.block_rows,
                      part=rank, n_parts=world, out_ptr=tile.data_ptr(), flags=flags, stats=stats)
            if world > 1:
                with torch.cuda.stream(serial_stream):      # same stream as the kernels: ordered after them
                    timed_gather(serial_stream, k, tile, tiling, rank, out=frame, scratch=scratch, via_host=(backend != "nccl"))
            return
        b = k % depth
        if world == 1:
            scenes[b].render(cam, W, H, n_shadow, max_depth, **common, block_rows=args.block_rows,
                             out_ptr=tiles[b].data_ptr(), flags=flags, stats=stats)
            return
        render_stream = render_streams[b]
        if tile_free[b] is not None:
            render_stream.wait_event(tile_free[b])          # the gather of frame k-depth has read this tile
        scenes[b].render(cam, W, H, n_shadow, max_depth, **common, block_rows=args.block_rows,
                         part=rank, n_parts=world, out_ptr=tiles[b].data_ptr(), flags=flags, stats=stats)
        rendered[b].record(render_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(rendered[b])
            timed_gather(comm_stream, k, tiles[b], tiling, rank, out=frame, scratch=scratch, via_host=(backend != "nccl"))
            ev = torch.cuda.Event(); ev.record(comm_stream); tile_free[b] = ev

    def barrier():
        if world > 1:
            dist.barrier()

    # setup: a few frames so that the library has grown its workspace and the GPU is at its sustained
    # clocks (kernel times settle after ~6 back-to-back frames, tools/frame_times.py), then the W warmup steps
    for _ in range(max(0, args.prewarm)):       # a fixed count: every rank must issue the same collectives
        step(_lib.Stats())
    for sc_ in scenes:
        sc_.sync()
    for _ in range(args.warmup):
        step(_lib.Stats())
    for sc_ in scenes:
        sc_.sync()
    frame_no[0] = 0
    gather_events.clear()
    all_stats = [_lib.Stats() for _ in range(args.steps)]
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in all_stats:
        step(s)
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    for sc_ in scenes:
        sc_.sync()   # folds counters / event timings into the rt_stats of the timed steps
    # N = 1 with two frames in flight: the same K frames once more, one after the other on ONE stream — the frame time
    # comparable with rounds 1-2, and the event-timed kernel durations of the roofline block (not distorted by overlap)
    timed_source, serial = all_stats, None
    if world == 1 and pipelined:
        ser_stats = [_lib.Stats() for _ in range(args.steps)]
        for _ in range(3):
            gs.render(cam, W, H, n_shadow, max_depth, **common, block_rows=args.block_rows,
                      out_ptr=tile.data_ptr(), flags=_lib.RT_FLAG_ASYNC)
        gs.sync(); torch.cuda.synchronize()
        ts = time.perf_counter()
        for k, s_ in enumerate(ser_stats):
            gs.render(cam, W, H, n_shadow, max_depth, **common, block_rows=args.block_rows, out_ptr=tile.data_ptr(),
                      flags=_lib.RT_FLAG_ASYNC | (_lib.RT_FLAG_TIMING if (not args.no_timing and k % every == 0) else 0), stats=s_)
        torch.cuda.synchronize()
        serial = (time.perf_counter() - ts) / args.steps * 1e3
        gs.sync()
        timed_source = ser_stats

    # whole-job numbers: max time over ranks, rays summed over ranks
    last = all_stats[-1]
    my = torch.tensor([elapsed, float(last.rays), float(last.hits), float(last.shadow_rays)], dtype=torch.float64,
                      device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = my[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = my[1:].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0]); rays, hits, shadow = (float(x) for x in sums)
    else:
        rays, hits, shadow = float(last.rays), float(last.hits), float(last.shadow_rays)

    # N > 1: what each rank spent computing (first kernel start -> last kernel end of its event-timed frames, rt_stats.total_ms)
    # and in the gather (events on the stream the gather runs on), so that the first hardware run can be read from its own
    # line: compute_ms.max ~ ms_per_step means the slowest rank's kernels bound the frame, gather_ms ~ ms_per_step the link
    multi = None
    if world > 1:
        timed_mine = [s_ for k, s_ in enumerate(all_stats) if k % every == 0] if not args.no_timing else []
        comp = sum(s_.total_ms for s_ in timed_mine) / max(len(timed_mine), 1)
        gath = sum(a.elapsed_time(b) for a, b in gather_events) / max(len(gather_events), 1)
        mine = torch.tensor([comp, gath, float(tiling.n_rows(rank))], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        every_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_rank, mine)
        per = [[float(x) for x in t] for t in every_rank]
        multi = {"compute_ms": {"per_rank": [round(p_[0], 4) for p_ in per], "max": round(max(p_[0] for p_ in per), 4), "min": round(min(p_[0] for p_ in per), 4)},
                 "gather_ms": {"per_rank": [round(p_[1], 4) for p_ in per], "rank0": round(per[0][1], 4)},
                 "rows_per_rank": [int(p_[2]) for p_ in per], "tile_rows_gathered": tiling.max_rows,
                 "frames_with_events": len(timed_mine), "pipelined": bool(pipelined), "frames_in_flight": depth,
                 "note": "compute_ms: first kernel start -> last kernel end of a rank's frame (HIP events on its render stream); gather_ms: "
                         "torch.distributed.gather of the padded tiles + reassembly on the stream it runs on (rank 0 waits for the slowest peer); "
                         "with k frames in flight the frames share the GPU: a compute-bound rank then shows compute_ms ~ k x ms_per_step, "
                         "RT_BENCH_PIPELINE=0 gives the unshared frame time",
                 "kernel_ms": {"per_rank_note": "sum of the event-timed kernels' own durations of a frame on rank 0 (trace + shade + other)",
                               "rank0": round(sum(s_.trace_ms + s_.shade_ms + s_.other_ms for s_ in timed_mine) / max(len(timed_mine), 1), 4)}}

    verify = None
    if not args.no_verify and world > 1 and rank == 0:
        full = torch.empty((H, W, 3), dtype=dtype, device=dev)
        torch.cuda.synchronize()
        gs.render(cam, W, H, n_shadow, max_depth, **common, out_ptr=full.data_ptr())
        torch.cuda.synchronize()
        verify = {"max_abs_diff_vs_single_gpu_frame": float((full.double() - frame.double()).abs().max()), "ranks": dist.get_world_size(), "backend": backend}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_rays = rays + shadow
        value = total_rays / (ms_per_step * 1e-3) / 1e6
        n_sph, n_pln, n_box = packed.counts
        # dominant kernel on this rank: shade launches of the timed steps
        timed = [s for k, s in enumerate(timed_source) if k % every == 0] if not args.no_timing else []
        shade_ms = sum(s.shade_ms for s in timed) / max(len(timed), 1)     # frames bracketed with HIP events
        trace_ms = sum(s.trace_ms for s in timed) / max(len(timed), 1)
        launches = max(timed_source[0].shade_launches, 1)
        seg = max(int(last.segments), 0)
        # (frames whose shade kernel traces its own rays — k_shade_ls, fused — execute those queries in the timed launches too)
        fused_rays, fused_children = rays_traced_by_shade_kernel(timed[0].as_dict()) if timed else (0, 0)
        flops_frame = shade_kernel_flops(n_sph=n_sph, n_pln=n_pln, n_box=n_box, n_lights=len(lights), n_shadow=n_shadow,
                                         hits=int(last.hits), segments=seg, rays=fused_rays, children=fused_children)
        roofline = None
        if shade_ms > 0:
            peak = PEAK_TFLOPS[precision]
            effective = flops_frame / (shade_ms * 1e-3) / 1e12
            small = n_pln <= 2 and ((n_box == 0 and n_sph <= 6) or (0 < n_box <= 4 and n_sph <= 4))
            kernel = (("k_shade_ls" if small else "k_shade_fast") if precision == "fp32" else "k_shade_exact")
            # fp32: the frame's shade kernels skip most of the flop model's work (conservative culling, fully-lit / umbra
            # shortcuts), so model flops / their time is an EFFECTIVE rate that can exceed the peak.  The roofline fraction
            # is therefore measured on frames rendered with RT_FLAG_BRUTE_FORCE (same kernels, every surface tested for every
            # shadow sample = the work the flop model counts), outside the timed region.
            brute = None
            if precision == "fp32" and world == 1 and args.brute_frames > 0:
                bs = []
                for _ in range(args.brute_frames + 1):
                    st_b = _lib.Stats()
                    gs.render(cam, W, H, n_shadow, max_depth, **common, out_ptr=tile.data_ptr(),
                              flags=_lib.RT_FLAG_TIMING | _lib.RT_FLAG_BRUTE_FORCE, stats=st_b)
                    bs.append(st_b)
                bs = bs[1:]                                   # first one: warm-up
                b_shade = sum(x.shade_ms for x in bs) / len(bs); b_total = sum(x.total_ms for x in bs) / len(bs)
                b_launches = max(bs[0].shade_launches, 1)
                b_ach = flops_frame / (b_shade * 1e-3) / 1e12
                brute = {"frames": len(bs), "frame_ms": round(b_total, 4), "shade_ms": round(b_shade, 4), "launch_ms": b_shade / b_launches,
                         "launches_per_frame": b_launches, "achieved": round(b_ach, 3), "frac": round(b_ach / peak, 4),
                         "rays_equal": int(bs[0].rays) == int(last.rays) and int(bs[0].hits) == int(last.hits)}
            traffic = None
            traffic_frame = None
            executed = None      # counted instructions / flops of the same launches from the committed PMC pass (profiles/)
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc):
                try:
                    pmc_all = json.load(open(pmc))
                    rec = pmc_all.get(args.workload, {})
                    rec_b = pmc_all.get(args.workload + "_brute", {})
                    if brute is not None and rec_b.get("valu_wave_instructions_per_frame") and rec_b.get("valu_issue_peak_Tinstr_per_s"):
                        # the brute-force launches' own counters (a separate rocprofv3 --pmc pass over quick_bench.py --brute)
                        brute["valu_wave_instructions_per_frame"] = rec_b["valu_wave_instructions_per_frame"]
                        brute["valu_issue_fraction_of_measured_peak"] = round(
                            rec_b["valu_wave_instructions_per_frame"] / (brute["shade_ms"] * 1e-3) / 1e12 / rec_b["valu_issue_peak_Tinstr_per_s"], 4)
                        brute["model_flops_per_valu_lane_instruction"] = round(flops_frame / (64.0 * rec_b["valu_wave_instructions_per_frame"]), 4)
                        brute["pmc_source"] = rec_b.get("source")
                    traffic = rec.get("hbm_bytes_per_shade_launch")
                    traffic_frame = rec.get("hbm_bytes_per_frame")
                    if world == 1 and precision == "fp64" and rec.get("executed_fp64_flops_per_frame"):
                        ex = rec["executed_fp64_flops_per_frame"] / (shade_ms * 1e-3) / 1e12
                        issue_peak = rec.get("valu_issue_peak_Tinstr_per_s")
                        executed = {"tflops": round(ex, 2), "frac": round(ex / peak, 4),
                                    "valu_wave_instructions_per_frame": rec.get("valu_wave_instructions_per_frame"),
                                    "valu_issue_fraction_of_measured_peak": round(rec["valu_wave_instructions_per_frame"] / (shade_ms * 1e-3) / 1e12 / issue_peak, 4)
                                    if issue_peak and rec.get("valu_wave_instructions_per_frame") else None,
                                    "valu_active_fraction_of_wave_cycles": rec.get("valu_active_fraction_of_wave_cycles"),
                                    "round": rec.get("round"), "source": rec.get("source", "profiles/pmc_latest.json (rocprofv3 --pmc)"),
                                    "note": "SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 x 64 lanes (FMA = 2) of the shade launches / their time"}
                    if world == 1 and precision == "fp32" and rec.get("executed_fp32_flops_per_frame"):
                        ex = rec["executed_fp32_flops_per_frame"] / (shade_ms * 1e-3) / 1e12
                        ex_hi = rec.get("executed_fp32_flops_per_frame_upper", rec["executed_fp32_flops_per_frame"]) / (shade_ms * 1e-3) / 1e12
                        issue_peak = rec.get("valu_issue_peak_Tinstr_per_s")
                        executed = {"tflops": round(ex, 2), "frac": round(ex / peak, 4),
                                    "tflops_if_all_packed": round(ex_hi, 2), "frac_if_all_packed": round(ex_hi / peak, 4),
                                    "packed_static_share_of_fp32_add_mul_fma": rec.get("packed_static_share_of_fp32_add_mul_fma"),
                                    "valu_wave_instructions_per_frame": rec.get("valu_wave_instructions_per_frame"),
                                    "valu_issue_fraction_of_measured_peak": round(rec["valu_wave_instructions_per_frame"] / (shade_ms * 1e-3) / 1e12 / issue_peak, 4)
                                    if issue_peak and rec.get("valu_wave_instructions_per_frame") else None,
                                    "valu_active_fraction_of_wave_cycles": rec.get("valu_active_fraction_of_wave_cycles"),
                                    "waves_per_simd": rec.get("waves_per_simd"), "round": rec.get("round"),
                                    "source": rec.get("source", "profiles/pmc_latest.json (rocprofv3 --pmc)")}
                except Exception:
                    traffic = None
            culled = {"shade_ms_per_step": round(shade_ms, 4), "launch_ms": shade_ms / launches, "launches_per_step": launches,
                      "effective_tflops": round(effective, 3), "effective_frac": round(effective / peak, 4),
                      "note": "model flops / time of the culled kernels: useful work per second, not utilisation (can exceed 1)"}
            if brute is not None:
                achieved, frac, basis = brute["achieved"], brute["frac"], "brute_force"
            elif precision == "fp64":
                # k_shade_exact's only skipped work is fan_cull's candidate sets; no brute-force switch exists for it
                achieved, frac, basis = round(effective, 3), round(effective / peak, 4), "culled (fp64 kernel: effective rate)"
            else:
                achieved, frac, basis = None, None, "not measured (N > 1 or --brute-frames 0)"
            roofline = {"bound": "valu", "kernel": kernel,
                        "flop_model": "brute-force algorithmic flops of SURVEY.md 8(d): every query tests every surface",
                        "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": frac, "frac_basis": basis,
                        "traffic": traffic, "traffic_per_frame": traffic_frame, "algorithmic_bytes_per_frame": int(W * H * 3 * (1 if u8 else 4 if precision == "fp32" else 8)),
                        "brute_force": brute, "culled": culled, "executed": executed,
                        "flops_per_launch": flops_frame / launches, "frames_with_events": len(timed),
                        "rays_traced_in_shade_kernel": fused_rays,
                        "trace_ms_per_step": round(trace_ms, 4), "flops_per_query": flops_per_query(n_sph, n_pln, n_box)}
            if frac is not None and precision == "fp32":
                assert frac <= 1.0, f"roofline.frac {frac} > 1: the brute-force run did not execute the model's work"
        out = {
            "metric": "Mrays/s incl. shadow+secondary rays, pool.txt 4K" if args.workload == "pool4k"
                      else f"Mrays/s incl. shadow+secondary rays, {args.workload}",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "f64",
            "data": "reference scene values (scenes/pool.txt etc. in-repo), pixel-keyed jitter seed 1234",
            "config": {"workload": desc, "rays_per_frame": int(total_rays), "radiance_rays": int(rays), "hits": int(hits),
                       "shadow_rays": int(shadow), "row_blocks": args.block_rows,
                       "parallelism": f"row-tiled x{world}" + (f", {backend} gather to rank 0" if world > 1 else "")
                                      + (f", {depth} frames in flight (a scene handle and a stream each; gather k overlaps render k+1)" if (pipelined and world > 1) else "")
                                      + (f", {depth} frames in flight (a scene handle and a stream each)" if (pipelined and world == 1) else ""),
                       "frames_in_flight": depth, "out_format": "u8" if u8 else ("f32" if precision == "fp32" else "f64")},
            "roofline": roofline,
        }
        if world > 1 and roofline is not None:
            roofline["traffic"] = None      # the PMC pass (profiles/) measured full-frame launches on one GPU
            roofline["traffic_per_frame"] = None
        if verify is not None:
            out["verify"] = verify
        if multi is not None:
            out["multi"] = multi
        if serial is not None:
            out["one_frame_at_a_time"] = {"ms_per_step": round(serial, 4), "value": round(total_rays / (serial * 1e-3) / 1e6, 2), "unit": "Mrays/s",
                                          "steps": args.steps,
                                          "note": "the same frames one after the other on one stream (RT_BENCH_PIPELINE=0 makes this the headline): "
                                                  "frame time as measured in rounds 1-2; roofline.culled comes from these frames' HIP events"}
        if world == 1 and not pipelined:
            # Throughput with TWO frames in flight (two scene handles on two streams, as the N > 1 runs do): the next frame's
            # level 0 fills the tails of this frame's small launches.  Reported beside the headline, which stays one frame
            # after the other on one stream (`value`, `ms_per_step`).  Outside the timed region.
            try:
                second = GpuScene(packed, device=local_rank)
                pair = [gs, second]                   # (each handle renders on its own private non-blocking stream)
                tiles2 = [tile, torch.empty_like(tile)]
                def two(k_frames):
                    for k in range(k_frames):
                        pair[k & 1].render(cam, W, H, n_shadow, max_depth, **common, block_rows=args.block_rows,
                                           out_ptr=tiles2[k & 1].data_ptr(), flags=_lib.RT_FLAG_ASYNC)
                two(8)
                for sc_ in pair:
                    sc_.sync()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                two(args.steps)
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t2) / args.steps * 1e3
                for sc_ in pair:
                    sc_.sync()
                second.close()
                out["two_frames_in_flight"] = {"ms_per_step": round(ms2, 4), "value": round(total_rays / (ms2 * 1e-3) / 1e6, 2), "unit": "Mrays/s",
                                               "steps": args.steps, "note": "frames alternate between two scene handles on two streams; not the headline"}
            except Exception as e:      # (diagnostic only)
                out["two_frames_in_flight"] = {"error": str(e)[:200]}
        if world == 1:
            # The drop-in boundary: render_hip(...) as the reference's CLI would call it — scene objects in, host image out
            # (PCIe-inclusive, so never `value`).  fp32 frames come back as float32 (pinned, pooled); second call onwards.
            from raytracer_amd.render import render_hip
            out_dt = np.float32 if precision == "fp32" else np.float64
            bt = []
            for _ in range(4):
                tb = time.perf_counter()
                himg = render_hip(cam, st, mats, surfs, lights, W, H, precision=precision, jitter=("pixel", 1234), verbose=False, out_dtype=out_dt)
                bt.append((time.perf_counter() - tb) * 1e3)
                del himg
            out["boundary_ms"] = round(min(bt[1:]), 4)
            out["boundary"] = {"call": f"render_hip(camera, scene_settings, materials, surfaces, lights, {W}, {H}, precision='{precision}', out_dtype={np.dtype(out_dt).name})",
                               "ms": [round(x, 3) for x in bt], "bytes_to_host": int(W * H * 3 * np.dtype(out_dt).itemsize),
                               "note": "wall time of the Python call: pack + cached device scene + frame rendered in 3 row bands, each copied to pinned host memory while the next renders"}
        if world == 1 and not args.no_cpu_baseline:
            base, parity = cpu_baseline(args.workload, gs, cam, st, mats, surfs, lights, W, H, precision)
            out["cpu_baseline"] = base
            out["parity_check"] = parity
        print(json.dumps(out), flush=True)

    for sc_ in scenes:
        sc_.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
