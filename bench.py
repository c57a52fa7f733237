#!/usr/bin/env python3
"""Benchmark of the rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A step is one frame of the hot path (camera rays -> closest hit -> shading -> bounces -> accumulate/pack) over
one synthetic camera/scene already resident in HBM.  The default workload is BASELINE.json's configs[1]: the
reference's default plane+sphere scene (src/bin/scene3.rs) at 3840x2160, 1 spp, max 5 bounces.  The metric is
Mrays/s, a ray being one trace_ray or march_ray invocation (primary, reflection, refraction, shadow).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Frames are independent units, so each rank
renders its own frames of the sequence with no collective in the data path ("weak": per-GPU work is fixed); the
timing is barrier + synchronize on both sides and the maximum over ranks.  `--shard rows` instead splits every
frame into row bands (one per rank), and `--shard samples` splits the samples of an accumulated frame and
all-reduces the accumulators once per step.

Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X: 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz (packed FP32 FMA)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E
CLOCK_PREWARM_S = 0.1   # untimed rendering before the warm-up steps (see main)

WORKLOADS = {
    # name: (scene, width, height, max_ray_bounces or None, accumulate, samples per step, default sharding, description)
    "scene3_4k": ("scene3", 3840, 2160, 5, False, 1, "frames", "reference default scene (plane + sphere, 2 lights, shadows), 3840x2160, 1 spp, max 5 bounces"),
    "scene1_4k": ("scene1", 3840, 2160, 5, False, 1, "frames", "all-primitives textured scene, 3840x2160, 1 spp, max 5 bounces"),
    "tex4_4k": ("tex4", 3840, 2160, 5, False, 1, "frames", "four textured spheres over a chess plane, 3840x2160, 1 spp, max 5 bounces"),
    "scene4_4k": ("scene4", 3840, 2160, None, False, 1, "frames", "ray-marching SDF scene, 3840x2160, 255 max march steps"),
    "scene2_4k": ("scene2", 3840, 2160, 5, True, 1, "frames", "diffuse emissive path tracing with accumulation, 3840x2160"),
    "spheres10k_8k": ("spheres10k", 7680, 4320, 5, True, 64, "samples", "10 000 diffuse spheres, 7680x4320, 64 spp accumulated per frame"),
}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 1000 frames (5 for the 64-spp workload)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 20 frames (1 for the 64-spp workload)")
    ap.add_argument("--workload", default="scene3_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default=None, choices=["frames", "rows", "samples"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    # Libraries loaded below write banners to file descriptor 1 (RCCL prints its version block there).  The one JSON line is
    # the only thing that may reach the real stdout: keep a private handle to it and point descriptor 1 at stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU; RT_BENCH_BACKEND=gloo rehearses the multi-process launch of the default (frames) sharding on a
    # box with fewer GPUs than ranks (ranks then share devices round-robin; gloo's collectives on device tensors are
    # not what the rows / samples modes are written for)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    # RT_BENCH_FORCE_DIST=1 makes a single rank go through the process group as well (its barrier and reductions), which
    # rehearses the RCCL calls of the multi-GPU launch on a one-GPU box
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    torch.cuda.set_device(device_index)

    pkg = graft.load_package()
    scene_name, width, height, bounces, accumulate, spp, default_shard, description = WORKLOADS[args.workload]
    # a 4K frame takes 0.1-1 ms: 1000 timed frames take at most a second
    if args.steps is None:
        args.steps = 1000 if spp == 1 else 5
    if args.warmup is None:
        args.warmup = 20 if spp == 1 else 1
    shard = args.shard or default_shard
    scene = pkg.Scene(scene_name)
    if bounces is not None:
        scene.desc.max_ray_bounces = bounces
    ctx = pkg.Context(device_index)
    ctx.load(scene, width, height)

    rgba = torch.zeros((height, width, 4), dtype=torch.uint8, device="cuda")
    accum = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    my_rows = pkg.shard_rows(height, rank, world)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def render(sample_index: int, rows=None, acc: bool = accumulate) -> None:
        # the timed region is bracketed by this script's own events, so the library's per-launch timing events are off
        ctx.render_device(frame_index=sample_index if acc else 1, accumulate=acc, seed=0x5EED, rows=rows,
                          d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=sh, no_timing=True)

    def step(i: int) -> None:
        if shard == "frames":
            # whole frames of a sequence go to different ranks: rank r renders frames r, r + world, ...; no collective
            for k in range(spp):
                render(1 + (i * world + rank) * spp + k)
        elif shard == "rows":
            # one frame per step, one row band per rank; the bands are all-gathered so that every rank holds the frame
            for k in range(spp):
                render(1 + i * spp + k, rows=my_rows if world > 1 else None)
            if world > 1:
                rgba.copy_(pkg.gather_row_bands(dist, rgba, height, rank, world))
        else:
            # the spp samples of one accumulated frame are split by sample index; a reduce-scatter leaves each rank with
            # the summed rows it owns, which it resolves (divide, clamp, pack); the RGBA8 bands are then all-gathered
            accum.zero_()
            for idx in pkg.shard_samples(spp, rank, world, 1 + i * spp):
                render(idx, acc=True)
            rb, re, band = pkg.reduce_scatter_accumulators(dist if world > 1 else None, accum, rank, world)
            if band.data_ptr() == accum.data_ptr() + rb * width * 16:
                # all-reduce fallback or single GPU: the band is a slice of the full accumulator
                ctx.resolve_device(spp, rows=(rb, re), d_accum=accum.data_ptr(), d_rgba8=rgba.data_ptr(), stream=sh)
                if world > 1:
                    rgba.copy_(pkg.gather_row_bands(dist, rgba, height, rank, world))
            else:
                # the band is its own tensor: resolve it as rows [0, re - rb) straight into this rank's rows of the frame
                ctx.resolve_device(spp, rows=(0, re - rb), d_accum=band.data_ptr(), d_rgba8=rgba.data_ptr() + rb * width * 4, stream=sh)
                pkg.gather_equal_bands(dist, rgba[rb:re].clone(), rgba)   # the send buffer must not alias the receive buffer

    def rays_of_step(i: int) -> int:
        """Rays this rank traces in step i (untimed pass: one stats read-back per render)."""
        total = 0
        if shard == "samples":
            ids = pkg.shard_samples(spp, rank, world, 1 + i * spp)
        elif shard == "frames":
            ids = [1 + (i * world + rank) * spp + k for k in range(spp)]
        else:
            ids = [1 + i * spp + k for k in range(spp)]
        for idx in ids:
            render(idx, rows=my_rows if (shard == "rows" and world > 1) else None, acc=accumulate or shard == "samples")
            total += ctx.stats()["rays_total"]
        return total

    # The GPU raises its clock over the first ~50 ms of sustained work (measured: a scene3 frame goes from 0.125 ms to a steady
    # 0.104 ms over the first 500 frames).  Render untimed frames for that long before the warm-up steps, so that the timed
    # steps see the steady clock whatever --warmup says.
    # (Time-based, so only where steps have no collective: ranks may render different numbers of pre-warm frames.)
    t_pre = time.perf_counter()
    while (shard == "frames" or world == 1) and time.perf_counter() - t_pre < CLOCK_PREWARM_S:
        for _ in range(8 if spp == 1 else 1):
            step(0)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    launches_per_step = len(pkg.shard_samples(spp, rank, world)) if shard == "samples" else spp
    # average launch duration of the dominant kernel on the launch stream (launches are back to back; at 1 spp without
    # collectives this is the whole step)
    kernel_ms = ev0.elapsed_time(ev1) / (args.steps * launches_per_step)
    # rays per step: deterministic scenes trace the same rays every frame; random modes vary by < 0.1 % between sample sets,
    # so the count of the first timed step's sample set stands for every step
    rays_rank = rays_of_step(args.warmup)

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    r = torch.tensor([float(rays_rank)], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    elapsed_max = float(t.item())
    rays_per_step_all = float(r.item())

    if rank == 0:
        # algorithmic FLOPs of one frame: event counts of a (slower, untimed) counting render times the per-event constants
        ctx.render_device(frame_index=1, accumulate=False, seed=0x5EED, rows=my_rows if (shard == "rows" and world > 1) else None,
                          collect_events=True, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=sh)
        st = ctx.stats()
        flops = pkg.algorithmic_flops(st)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        value = rays_per_step_all * args.steps / elapsed_max / 1e6
        out = {
            "metric": "Mrays/sec (primary+secondary) and frame ms at 4K",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak" if shard == "frames" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {description}", "width": width, "height": height,
                       "max_ray_bounces": int(scene.desc.max_ray_bounces), "samples_per_step": spp,
                       "rays_per_step_per_rank": int(rays_rank), "sharding": shard, "clock_prewarm_ms": int(CLOCK_PREWARM_S * 1e3),
                       "parallelism": f"{shard}-sharded x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 5), "traffic": measured_traffic(args.workload),
                         # the 157.3 TFLOP/s peak is one FMA (2 flops) per lane every 2 cycles; bit-exact parity keeps multiplies
                         # and adds unfused, so one flop per lane-instruction (half of it) bounds this code
                         "frac_of_unfused_peak": round(achieved / (FP32_VALU_PEAK_TFLOPS / 2.0), 5),
                         "kernel": "rt_march_kernel" if scene.desc.ray_marching else "rt_trace_kernel",
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_flops_per_launch": int(flops),
                         "note": "FP32 vector (VALU) roofline: this path has no dense contraction, so no MFMA"},
        }
        traffic = out["roofline"]["traffic"]
        if traffic:
            # the secondary bound: measured HBM bytes of one launch (direction table read, accumulator and RGBA8 written) over
            # the launch time, against the 8 TB/s HBM3E peak
            gbps = traffic / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["hbm_view"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4)}
        if scene.desc.ray_marching:
            # one sdfs() call evaluates the whole SDF list once (src/ray_marching/ray_marching.rs:24-107)
            out["config"]["sdf_list_evals_per_s"] = round(st["sdfs_call"] / (kernel_ms * 1e-3), 1)
        if world == 1 and not args.no_cpu_baseline:   # a reported baseline of the single-GPU line only
            out["cpu_baseline"] = cpu_baseline(pkg, scene, width, height, accumulate)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(workload: str):
    """HBM bytes per launch of the workload's render kernel from the committed PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 runs, read side doubled for gfx950), or None when the workload has not been profiled."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(workload, {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_baseline(pkg, scene, width, height, accumulate) -> dict:
    """The CPU restatement of the reference algorithm (oracle/, the reference's 200-chunk decomposition over worker
    threads) timed on this box's host cores on a bounded sample of the same workload."""
    orc = graft.load_oracle()
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    cam = scene.camera(width, height)
    tex = scene.textures()
    # Camera.ray_directions is computed once per camera change in the reference (src/camera.rs:128-157), outside the frames
    dirs = orc.ray_directions(cam)
    # bounded sample: evenly spaced 8-row bands covering ~1/6 of the frame (all of it for cheap scenes), about 10-30 s of core time
    budget_rows = height if not scene.desc.ray_marching and scene.desc.n_objects < 100 else max(8, height // 48)
    n_bands = max(1, budget_rows // 8)
    starts = [int(round(i * (height - 8) / max(1, n_bands - 1))) // 8 * 8 for i in range(n_bands)] if n_bands * 8 < height else None
    rays = 0
    t0 = time.perf_counter()
    if starts is None:
        # cheap scenes: whole frames, repeated until the cores have worked for about 20 core-seconds
        orc.render(scene.desc, cam, tex, accumulate=accumulate, n_threads=threads, ray_dirs=dirs)   # warm-up (page faults, thread start)
        t0 = time.perf_counter()
        frames = 0
        while frames < 3 or (time.perf_counter() - t0) * threads < 20.0:
            r = orc.render(scene.desc, cam, tex, accumulate=accumulate, n_threads=threads, want_counters=True, ray_dirs=dirs)
            rays += sum(r["counters"][:5])
            frames += 1
        sample = f"{frames} full {width}x{height} frames"
    else:
        for y in sorted(set(starts)):
            r = orc.render(scene.desc, cam, tex, accumulate=accumulate, n_threads=threads, want_counters=True, rows=(y, y + 8), ray_dirs=dirs)
            rays += sum(r["counters"][:5])
        sample = f"{len(set(starts))} evenly spaced 8-row bands of the {width}x{height} frame"
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": sample + f", {rays} rays in {dt:.2f} s; CPU restatement of the reference (Rust toolchain unavailable)"}


if __name__ == "__main__":
    main()
