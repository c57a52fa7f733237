#!/usr/bin/env python3
"""Benchmark of the rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--shard rows|samples|frames]

A step is one frame of the hot path (camera rays -> closest hit -> shading -> bounces -> accumulate/pack) over
one synthetic camera/scene already resident in HBM.  The default workload is BASELINE.json's configs[1]: the
reference's default plane+sphere scene (src/bin/scene3.rs) at 3840x2160, 1 spp, max 5 bounces.  The metric is
Mrays/s, a ray being one trace_ray or march_ray invocation (primary, reflection, refraction, shadow).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Started without WORLD_SIZE in the environment,
this script starts the N ranks itself, as fresh child processes, before anything touches a GPU; under
torch.distributed.run it is one of the ranks.  The frame is split the way the reference splits it into chunks
(src/renderer.rs:104-143):
  rows     (1-spp workloads) every rank renders every N-th 8-row tile row of each frame; one all-gather of RGBA8 tile
           rows per frame leaves the whole frame on every rank.
  samples  (the 64-spp workload) every rank renders its share of the sample indices into its own accumulator; one
           reduce-scatter(sum) of the accumulators, resolve of the owned rows, one all-gather of RGBA8 bands per frame.
  frames   (opt-in) whole frames of a sequence per rank, no collective.
The timing is barrier + synchronize on both sides and the maximum over ranks.  Rank 0 prints one JSON line.

Without --workload the line is the default workload's, and three short measurements ride along in the same line
(--no-riders leaves them out): under "config5" BASELINE.json's multi-GPU configuration -- the 8K / 10 000-sphere / 64-spp
frame, sharded by sample index when N > 1 -- so that one run per GPU count tells how that frame scales as well; and, at
N = 1, under "changed_frames" the frames the reference renders when something changed (src/app.rs:214-248: every frame it
renders without accumulation): the default scene with a new camera pose every step (rt_set_camera + render) and the
all-primitives scene with its update function's object motion every step (rt_update_objects + render).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X: 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz (packed FP32 FMA)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E
CLOCK_PREWARM_S = 0.1   # untimed rendering before the warm-up steps (see main)

WORKLOADS = {
    # name: (scene, width, height, max_ray_bounces or None, accumulate, samples per step, sharding for N > 1, description)
    "scene3_4k": ("scene3", 3840, 2160, 5, False, 1, "rows", "reference default scene (plane + sphere, 2 lights, shadows), 3840x2160, 1 spp, max 5 bounces"),
    "scene1_4k": ("scene1", 3840, 2160, 5, False, 1, "rows", "all-primitives textured scene, 3840x2160, 1 spp, max 5 bounces"),
    "tex4_4k": ("tex4", 3840, 2160, 5, False, 1, "rows", "four textured spheres over a chess plane, 3840x2160, 1 spp, max 5 bounces"),
    "scene4_4k": ("scene4", 3840, 2160, None, False, 1, "rows", "ray-marching SDF scene, 3840x2160, 255 max march steps"),
    "scene2_4k": ("scene2", 3840, 2160, 5, True, 1, "rows", "diffuse emissive path tracing with accumulation, 3840x2160"),
    "spheres10k_8k": ("spheres10k", 7680, 4320, 5, True, 64, "samples", "10 000 diffuse spheres, 7680x4320, 64 spp accumulated per frame"),
    "spheres10k_4k": ("spheres10k", 3840, 2160, 5, True, 1, "rows", "10 000 diffuse spheres, 3840x2160, 1 spp per step (accumulating)"),
    "scene3_800x600": ("scene3", 800, 600, None, False, 1, "rows", "reference default scene at its default viewport (800x600, max 4 bounces): BASELINE config 1"),
    # frames in which something changed since the previous one (the only frames the reference renders without accumulation)
    "scene3_4k_moving": ("scene3", 3840, 2160, 5, False, 1, "rows", "reference default scene, 3840x2160, 1 spp, max 5 bounces, a new camera pose every step (rt_set_camera + render)"),
    "scene1_4k_animated": ("scene1", 3840, 2160, 5, False, 1, "rows", "all-primitives textured scene, 3840x2160, 1 spp, max 5 bounces, its cuboid turned every step (rt_update_objects + render)"),
}
MOTION = {"scene3_4k_moving": "camera", "scene1_4k_animated": "objects"}
N_POSES = 64   # camera poses / object states a moving workload cycles through


def spawn_ranks(n: int) -> int:
    """Starts the N ranks of a single-node run as fresh child processes of this (GPU-free) parent and waits for them.
    Rank 0 inherits stdout, so its JSON line is this command's output.  Returns the worst exit code."""
    import random
    import socket
    port = 0
    for _ in range(64):   # a rendezvous port below the kernel's range for outgoing connections, which nothing grabs in between
        cand = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", cand))
            port = cand
            break
        except OSError:
            continue
        finally:
            s.close()
    if port == 0:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 1000 frames (5 for the 64-spp workload)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 20 frames (1 for the 64-spp workload)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: scene3_4k (BASELINE configs[1]), followed by a short config-5 measurement")
    ap.add_argument("--shard", default=None, choices=["frames", "rows", "samples"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--consecutive-launches", action="store_true",
                    help="multi-sample workloads: one rt_render_device call per sample instead of one rt_render_samples_device call per frame (profiling runs)")
    ap.add_argument("--no-config5", "--no-riders", dest="no_config5", action="store_true",
                    help="default workload only: without the ride-along measurements (8K / 10 000-sphere / 64-spp frame; moving camera; animated scene)")
    ap.add_argument("--keep-accum-store", action="store_true",
                    help="non-accumulating workloads: also write the frame's linear value to the accumulator (default: RT_RENDER_SKIP_ACCUM_STORE)")
    ap.add_argument("--march-all-entries", action="store_true",
                    help="ray-marching scenes: every march step evaluates every entry of Scene.sdfs (RT_RENDER_MARCH_ALL_ENTRIES)")
    ap.add_argument("--no-sky-tiles", action="store_true",
                    help="small scenes: render without the per-tile classification that lets workgroups of sky end without tracing (RT_RENDER_NO_SKY_TILES)")
    ap.add_argument("--no-first-hit-reuse", action="store_true",
                    help="large diffuse scenes, single frames: trace every frame's camera rays instead of reading the hits an earlier frame of the same camera found (RT_RENDER_NO_FIRST_HIT_REUSE)")
    args = ap.parse_args()
    with_config5 = args.workload is None and not args.no_config5 and args.shard in (None, "samples")
    if args.workload is None:
        args.workload = "scene3_4k"
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # nothing in this process has touched a GPU (or imported torch) yet: the ranks are fresh processes
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or leave WORLD_SIZE unset and let bench.py start them)")
    # Libraries loaded below write banners to file descriptor 1 (RCCL prints its version block there).  The one JSON line is
    # the only thing that may reach the real stdout: keep a private handle to it and point descriptor 1 at stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL.  RT_BENCH_BACKEND=gloo rehearses the multi-process launch on a box with fewer GPUs than
    # ranks: ranks then share devices round-robin and the collectives are staged through host memory.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        sys.exit(f"bench.py: {world} ranks but {n_dev} GPU(s) visible; RT_BENCH_BACKEND=gloo rehearses the launch on fewer GPUs")
    device_index = local_rank if backend == "nccl" else local_rank % max(1, n_dev)
    # RT_BENCH_FORCE_DIST=1 makes a single rank go through the process group as well, which rehearses the RCCL calls of
    # the multi-GPU launch on a one-GPU box
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    dist = None
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

    def measure(workload: str, steps, warmup, want_cpu_baseline: bool, shard_arg):
        """One workload through warm-up and K timed steps; rank 0 gets the result line as a dict, the others None."""
        scene_name, width, height, bounces, accumulate, spp, multi_shard, description = WORKLOADS[workload]
        # a 4K frame takes 0.1-1 ms: 1000 timed frames take at most a second
        if steps is None:
            steps = 1000 if spp == 1 else 5
        if warmup is None:
            warmup = 20 if spp == 1 else 1
        shard = shard_arg or (multi_shard if use_dist else "none")
        if shard == "samples" and spp == 1:
            sys.exit("bench.py: --shard samples needs a multi-sample workload (spheres10k_8k)")
        scene = pkg.Scene(scene_name)
        if bounces is not None:
            scene.desc.max_ray_bounces = bounces
        ctx = pkg.Context(device_index)
        ctx.load(scene, width, height)
        # workloads in which something changes before every frame: the states are built once, a step only hands one over
        motion = MOTION.get(workload)
        poses, moved = [], []
        if motion == "camera":
            import math
            for k in range(N_POSES):   # a small orbit around the scene's own camera (src/bin/scene3.rs: position (0,2,5), direction (0,0,-1))
                a = 2.0 * math.pi * k / N_POSES
                poses.append(pkg.make_camera((0.25 * math.sin(a), 2.0 + 0.1 * math.cos(a), 5.0), (0.05 * math.sin(a + 1.0), 0.02 * math.cos(a), -1.0), width, height))
        elif motion == "objects":
            import ctypes
            for k in range(N_POSES):   # the scene's update function turns its cuboid, object 3 (src/bin/scene1.rs:13-24)
                o = pkg.rt_object.from_buffer_copy(scene.desc.objects[3])
                o.rotation_axis[1] += 360.0 * k / N_POSES
                pkg.host().rth_object_update(ctypes.byref(o))
                moved.append((pkg.rt_object * 1)(o))

        def apply_motion(i: int) -> None:
            if motion == "camera":
                ctx.set_camera(poses[i % N_POSES])
            elif motion == "objects":
                ctx.update_objects_array(3, moved[i % N_POSES])

        # every buffer a step touches exists before the timed loop: frame, accumulator, and the collectives' send/receive blocks
        xch = pkg.FrameExchange(dist, height, width, rank, world, device="cuda", backend=backend, force_collectives=use_dist)
        rgba, accum = xch.frame, xch.accum
        stream = torch.cuda.current_stream()
        sh = stream.cuda_stream
        rows_arg = xch.render_rows() if (shard == "rows" and xch.collectives) else None

        def barrier():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        # Non-accumulating frames: nothing reads Renderer.accumulated (src/renderer.rs:67-68 writes it, no caller of render() reads
        # it), so the frame's linear value is not written to it: 4 instead of 20 bytes stored per pixel.  The parity tests keep it.
        skip_store = not args.keep_accum_store

        def render(sample_index: int, acc: bool = accumulate, rows=rows_arg, events: bool = False, skip: bool = None) -> None:
            # the timed region is bracketed by this script's own events, so the library's per-launch timing events are off
            skip = skip_store if skip is None else skip
            ctx.render_device(frame_index=sample_index if acc else 1, accumulate=acc, seed=0x5EED,
                              flags=(pkg.RT_RENDER_SKIP_ACCUM_STORE if (skip and not acc) else 0) | (pkg.RT_RENDER_NO_SKY_TILES if args.no_sky_tiles else 0) |
                                    (pkg.RT_RENDER_MARCH_ALL_ENTRIES if args.march_all_entries else 0) |
                                    (pkg.RT_RENDER_NO_FIRST_HIT_REUSE if args.no_first_hit_reuse else 0),
                              rows=(rows[0], rows[1]) if rows else None, tile_row_stride=rows[2] if rows else 0,
                              d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=sh, no_timing=not events, collect_events=events)

        def sample_ids(i: int):
            if shard == "samples":
                return pkg.shard_samples(spp, rank, world, 1 + i * spp)
            if shard == "frames":
                return [1 + (i * world + rank) * spp + k for k in range(spp)]
            return [1 + i * spp + k for k in range(spp)]

        kernel_events, exchange_events = [], []
        # multi-sample frames go through rt_render_samples_device (one call per frame and rank; sample_ids are contiguous)
        batched = spp > 1 and not args.consecutive_launches and (shard == "samples" or accumulate) and shard != "frames"

        def step(i: int, timed: bool = False) -> None:
            # per-step events only where a step has collectives to separate from the render launches (an event pair costs
            # several microseconds on the stream: too much for a 0.1 ms frame); otherwise one pair brackets the timed loop
            timed = timed and xch.collectives and shard in ("rows", "samples")
            if timed:
                k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                k0.record(stream)
            if shard == "samples":
                accum.zero_()
            apply_motion(i)
            ids = sample_ids(i)
            if batched and len(ids) > 1:
                # the samples of a frame in one call (large diffuse scenes: one launch in which every pixel's samples follow each other)
                ctx.render_samples_device(ids[0], len(ids), seed=0x5EED, rows=(rows_arg[0], rows_arg[1]) if rows_arg else None,
                                          tile_row_stride=rows_arg[2] if rows_arg else 0, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=sh)
            else:
                for idx in ids:
                    render(idx, acc=True if shard == "samples" else accumulate)
            if timed:
                k1.record(stream)
                kernel_events.append((k0, k1))
            if shard in ("none", "frames") or not xch.collectives:
                return
            if shard == "rows":
                xch.gather_interleaved()
            else:
                # a reduce-scatter leaves each rank with the summed rows it owns; it resolves them (divide, clamp, pack) and the
                # RGBA8 bands are all-gathered
                rb, re, band = xch.sum_accumulators()
                if xch.reduce_scatter:
                    ctx.resolve_device(spp, rows=(0, re - rb), d_accum=band.data_ptr(), d_rgba8=xch.band_rgba.data_ptr(), stream=sh)
                else:
                    ctx.resolve_device(spp, rows=(rb, re), d_accum=accum.data_ptr(), d_rgba8=rgba.data_ptr(), stream=sh)
                xch.gather_bands(rb, re)
            if timed:
                x1 = torch.cuda.Event(enable_timing=True)
                x1.record(stream)
                exchange_events.append((k1, x1))

        # The GPU raises its clock over the first ~50 ms of sustained work (measured: a scene3 frame goes from 0.125 ms to a steady
        # 0.104 ms over the first 500 frames).  Render untimed frames for that long before the warm-up steps, so that the timed
        # steps see the steady clock whatever --warmup says.  With collectives in the step every rank must run the same number of
        # steps, so there the pre-warm is a fixed count.
        if not xch.collectives or shard in ("none", "frames"):
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < CLOCK_PREWARM_S:
                for _ in range(8 if spp == 1 else 1):
                    step(0)
                torch.cuda.synchronize()
        else:
            for _ in range(200 if spp == 1 else 1):
                step(0)
            torch.cuda.synchronize()
        for i in range(warmup):
            step(i)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(steps):
            step(warmup + i, timed=True)
        ev1.record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        launches_per_step = len(sample_ids(0))
        # average launch duration of the render kernel(s), from events on the launch stream: around each step's render launches
        # when the step also has collectives, around the whole timed loop (launches back to back) otherwise
        chained = batched and kernel_name(pkg, scene, spp) == "rt_wf_chain_kernel"
        if chained:
            # one launch renders all samples of the frame: its duration is the step's (timed once more here, alone on the stream)
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ids = sample_ids(warmup)
            p0.record(stream)
            ctx.render_samples_device(ids[0], len(ids), seed=0x5EED, rows=(rows_arg[0], rows_arg[1]) if rows_arg else None,
                                      tile_row_stride=rows_arg[2] if rows_arg else 0, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=sh)
            p1.record(stream)
            torch.cuda.synchronize()
            kernel_ms = p0.elapsed_time(p1)
            launches_per_step = 1
        elif batched:
            probe = min(8, launches_per_step)
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            render(1, acc=True)
            p0.record(stream)
            for k in range(probe):
                render(2 + k, acc=True)
            p1.record(stream)
            torch.cuda.synchronize()
            kernel_ms = p0.elapsed_time(p1) / probe
        elif kernel_events:
            kernel_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / (steps * launches_per_step)
        else:
            kernel_ms = ev0.elapsed_time(ev1) / (steps * launches_per_step)
        exchange_ms = sum(a.elapsed_time(b) for a, b in exchange_events) / steps if exchange_events else 0.0
        # Rays of the timed steps (untimed passes: one stats read-back per render).  Deterministic scenes trace the same rays
        # every frame; random modes vary by < 0.1 % between sample sets, so the first timed step's set stands for every step;
        # moving workloads are counted state by state.  rays_sky_tiles: camera rays that the tile classification settled
        # (workgroups of nothing but sky) -- the reference traces them, this backend does not, so they are reported apart.
        def count_step(i: int):
            apply_motion(i)
            total = sky = 0
            ids = sample_ids(i)
            if chained:
                ctx.render_samples_device(ids[0], len(ids), seed=0x5EED, rows=(rows_arg[0], rows_arg[1]) if rows_arg else None,
                                          tile_row_stride=rows_arg[2] if rows_arg else 0, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=sh)
                st = ctx.stats()
                return st["rays_total"], st["rays_reused_camera"] + st["rays_sky_tiles"]
            for idx in ids:
                render(idx, acc=True if shard == "samples" else accumulate)
                st = ctx.stats()
                total += st["rays_total"]
                sky += st["rays_sky_tiles"] + st["rays_reused_camera"]   # neither is traced here; the reference traces both
            return total, sky
        if motion:
            per_state = [count_step(k) for k in range(N_POSES)]
            rays_timed = sum(per_state[(warmup + i) % N_POSES][0] for i in range(steps))
            sky_timed = sum(per_state[(warmup + i) % N_POSES][1] for i in range(steps))
        else:
            r1, s1 = count_step(warmup)
            rays_timed, sky_timed = r1 * steps, s1 * steps
        rays_rank = rays_timed // steps
        # launches of the render kernel issued after the timed steps (tools/summarize_profile.py picks the timed ones out of a
        # kernel trace by counting back from the end)
        launches_after = (N_POSES if motion else 1) * (1 if chained else len(sample_ids(warmup))) + (1 if chained else (1 + min(8, launches_per_step) if batched else 0))

        # the same frames with the reference's write of the linear value to Renderer.accumulated (src/renderer.rs:68) left in
        ms_with_store = None
        if not accumulate and shard == "none" and skip_store:
            k2 = max(1, min(steps, 300))
            for i in range(10):
                apply_motion(i); render(1, skip=False)
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            q0.record(stream)
            for i in range(k2):
                apply_motion(i); render(1, skip=False)
            q1.record(stream)
            torch.cuda.synchronize()
            ms_with_store = q0.elapsed_time(q1) / k2
            launches_after += 10 + k2

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        r = torch.tensor([float(rays_timed), float(sky_timed)], dtype=torch.float64, device="cuda")
        if use_dist:
            if xch.staged:
                t, r = t.cpu(), r.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
        elapsed_max = float(t.item())
        rays_all, sky_all = float(r[0].item()), float(r[1].item())

        out = None
        if rank == 0:
            # algorithmic FLOPs of one launch: event counts of a (slower, untimed) counting render times the per-event constants
            render(1, acc=False, events=True)
            st = ctx.stats()
            flops = pkg.algorithmic_flops(st) * (spp if chained else 1)   # a launch of the chain kernel renders all samples of the frame
            achieved = flops / (kernel_ms * 1e-3) / 1e12
            value = rays_all / elapsed_max / 1e6
            value_traced = (rays_all - sky_all) / elapsed_max / 1e6
            traffic, traffic_note = measured_traffic(workload, pkg.LIB_PATH)
            if world > 1 and shard != "frames":   # the committed bytes are those of a launch over the whole frame on one GPU
                traffic, traffic_note = None, "PMC traffic is measured for the one-GPU launch of this workload; a rank's launch renders a share of it"
            hinfo = ctx.hierarchy_info()
            if args.no_sky_tiles:
                sky_tiles = "off (--no-sky-tiles)"
            elif hinfo["classified_workgroups"]:
                sky_tiles = (f"{hinfo['sky_workgroups']} of {hinfo['classified_workgroups']} workgroups hold only tiles that no camera ray leaves towards an object; "
                             "they write the value of a miss without tracing (classified when scene, camera or rows change: here once, before the timed steps)")
            else:
                sky_tiles = "not used by this workload's kernel"
            out = {
                "metric": "Mrays/sec (primary+secondary) and frame ms at 4K",
                "value": round(value, 2),
                # the same without the rays that were settled without tracing them: camera rays of all-sky workgroups (tile
                # classification) and camera rays whose hit an earlier sample of the launch or an earlier frame of the same
                # camera and scene had found (large diffuse scenes: traced once per pixel)
                "value_traced_only": round(value_traced, 2),
                "unit": "Mrays/s",
                "n_gpus": world,
                "steps": steps,
                "warmup": warmup,
                "ms_per_step": round(elapsed_max / steps * 1e3, 4),
                "higher_is_better": True,
                "scaling": "weak" if shard == "frames" else "strong",
                "vs_baseline": None,
                "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": f"{workload}: {description}", "width": width, "height": height,
                           "max_ray_bounces": int(scene.desc.max_ray_bounces), "samples_per_step": spp,
                           "rays_per_step_per_rank": int(rays_rank), "traced_rays_per_step": int(round((rays_all - sky_all) / steps)),
                           "untraced_rays_per_step": int(round(sky_all / steps)), "sharding": shard, "clock_prewarm_ms": int(CLOCK_PREWARM_S * 1e3),
                           "accumulator_store": "n/a (accumulating)" if accumulate or shard == "samples" else ("skipped" if skip_store else "written"),
                           "sky_tiles": sky_tiles, "render_launches_after_timed_steps": launches_after,
                           "ranks": world, "backend": backend if use_dist else "none", "exchange_ms_per_step": round(exchange_ms, 4),
                           "collective": ("none" if shard in ("none", "frames") or not xch.collectives else
                                          "all_gather(rgba8 tile rows)" if shard == "rows" else
                                          ("reduce_scatter(f32 accum) + all_gather(rgba8 bands)" if xch.reduce_scatter else "all_reduce(f32 accum) + all_gather(rgba8 bands)")),
                           "parallelism": f"{shard}-sharded x{world}" if world > 1 else "single GPU",
                           "samples_call": ("rt_render_samples_device (one launch per frame: every pixel's samples are traced one after the other by the "
                                            "wavefront that holds the pixel; roofline.kernel_ms is that launch)") if chained else
                                           ("rt_render_samples_device" if batched else "rt_render_device per sample")},
                "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 5), "traffic": traffic,
                             # the 157.3 TFLOP/s peak is one FMA (2 flops) per lane every 2 cycles; bit-exact parity keeps multiplies
                             # and adds unfused, so one flop per lane-instruction (half of it) bounds this code
                             "frac_of_unfused_peak": round(achieved / (FP32_VALU_PEAK_TFLOPS / 2.0), 5),
                             "kernel": kernel_name(pkg, scene, spp if batched else 1),
                             "kernel_ms": round(kernel_ms, 4), "algorithmic_flops_per_launch": int(flops),
                             "note": "FP32 vector (VALU) roofline: this path has no dense contraction, so no MFMA"},
            }
            if traffic_note:
                out["roofline"]["traffic_note"] = traffic_note
            if traffic:
                # the secondary bound: measured HBM bytes of one launch over the launch time, against the 8 TB/s HBM3E peak, and
                # against the bytes the frame has to move at the least (4 B RGBA8 per pixel; 36 B when accumulating)
                gbps = traffic / (kernel_ms * 1e-3) / 1e9
                alg = width * height * (36 if (accumulate or shard == "samples") else 4)
                out["roofline"]["hbm_view"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                                               "algorithmic_bytes": alg, "traffic_over_algorithmic": round(traffic / alg, 2)}
            if ms_with_store is not None:
                out["config"]["ms_per_step_with_accumulator_store"] = round(ms_with_store, 4)
            if motion:
                out["config"]["per_step_host_calls"] = "rt_set_camera + rt_render_device" if motion == "camera" else "rt_update_objects (1 object) + rt_render_device"
            if scene.desc.ray_marching:
                # one sdfs() call evaluates the whole SDF list once (src/ray_marching/ray_marching.rs:24-107)
                out["config"]["sdf_list_evals_per_s"] = round(st["sdfs_call"] / (kernel_ms * 1e-3), 1)
            if world == 1 and want_cpu_baseline:   # a reported baseline of the single-GPU line only
                cb = cpu_baseline(pkg, scene, width, height, accumulate)
                out["cpu_baseline"] = cb
                # the north star's ratio with its spread: all rays of the frame (what the CPU traces) and traced rays only
                out["vs_cpu_baseline"] = {"all_rays": {"median": round(value / cb["value"], 1), "min": round(value / cb["max"], 1), "max": round(value / cb["min"], 1)},
                                          "traced_rays_only": {"median": round(value_traced / cb["value"], 1), "min": round(value_traced / cb["max"], 1),
                                                               "max": round(value_traced / cb["min"], 1)}}
        ctx.close()
        del xch, rgba, accum
        torch.cuda.empty_cache()
        return out

    out = measure(args.workload, args.steps, args.warmup, not args.no_cpu_baseline, args.shard)
    if with_config5 and world == 1 and not use_dist:
        # the frames the reference renders when something changed (camera pose / object motion before every frame)
        changed = {}
        for wl, k in (("scene3_4k_moving", 500), ("scene1_4k_animated", 300)):
            extra = measure(wl, k, 20, False, None)
            if extra is not None:
                changed[wl] = {"workload": extra["config"]["workload"], "ms_per_step": extra["ms_per_step"], "value": extra["value"],
                               "value_traced_only": extra["value_traced_only"], "unit": extra["unit"], "steps": extra["steps"],
                               "ms_per_step_with_accumulator_store": extra["config"].get("ms_per_step_with_accumulator_store"),
                               "per_step_host_calls": extra["config"]["per_step_host_calls"], "kernel_ms": extra["roofline"]["kernel_ms"],
                               "frac": extra["roofline"]["frac"]}
        if out is not None:
            out["changed_frames"] = changed
    if with_config5:
        # BASELINE's multi-GPU configuration (8K, 10 000 spheres, 64 spp, sharded by sample index) rides along as a short
        # second measurement, so that a run at N GPUs of the default workload also says how that frame scales
        extra = measure("spheres10k_8k", 2, 1, False, None)
        if out is not None and extra is not None:
            out["config5"] = {"workload": extra["config"]["workload"], "value": extra["value"], "unit": extra["unit"], "n_gpus": extra["n_gpus"],
                              "steps": extra["steps"], "warmup": extra["warmup"], "ms_per_step": extra["ms_per_step"], "scaling": extra["scaling"],
                              "sharding": extra["config"]["sharding"], "collective": extra["config"]["collective"],
                              "exchange_ms_per_step": extra["config"]["exchange_ms_per_step"],
                              "kernel": extra["roofline"]["kernel"], "kernel_ms": extra["roofline"]["kernel_ms"], "frac": extra["roofline"]["frac"]}
    if out is not None:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def kernel_name(pkg, scene, samples: int = 1) -> str:
    """The kernel rt_render_device (samples == 1) or rt_render_samples_device launches for this scene (the choice in
    ray-tracing_amd/csrc/rt_mi355x.hip)."""
    d = scene.desc
    if d.ray_marching:
        return "rt_march_kernel"
    refractive = any(d.materials[i].kind == pkg.RT_REFRACTIVE for i in range(d.n_materials))
    bounded = sum(1 for i in range(d.n_objects) if d.objects[i].kind not in (pkg.RT_PLANE, pkg.RT_UNION, pkg.RT_SUBSTRACTION))
    if d.diffuse and not refractive and d.max_ray_bounces >= 1:
        if bounded >= 32:
            return "rt_wf_chain_kernel" if samples > 1 else "rt_wf_diffuse_kernel"
        return "rt_refill_diffuse_kernel"
    return "rt_trace_kernel"


def file_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def measured_traffic(workload: str, lib_path: str):
    """(HBM bytes per launch, note).  The bytes come from the committed PMC passes (FETCH_SIZE and WRITE_SIZE in separate
    rocprofv3 runs, read side doubled for gfx950; tools/profile.sh + tools/update_hbm_traffic.py refresh them) and are
    only reported when they were measured on the very library file that is running now (its sha256 is stored with them):
    otherwise None and a note, because kernels change and a stale constant would pass for a measurement."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            entry = json.load(fh).get(workload)
    except (OSError, ValueError):
        entry = None
    if not entry:
        return None, "no PMC traffic measurement committed for this workload"
    if entry.get("lib_sha256") != file_sha256(lib_path):
        return None, "the committed PMC traffic was measured on a different build of librt_mi355x.so (tools/profile.sh refreshes it)"
    return entry.get("bytes_per_launch"), None


def cpu_baseline(pkg, scene, width, height, accumulate) -> dict:
    """The CPU restatement of the reference algorithm (oracle/, the reference's 200-chunk decomposition over worker
    threads) timed on this box's host cores on a bounded sample of the same workload."""
    orc = graft.load_oracle()
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    cam = scene.camera(width, height)
    tex = scene.textures()
    # Camera.ray_directions is computed once per camera change in the reference (src/camera.rs:128-157), outside the frames
    dirs = orc.ray_directions(cam)
    # Bounded sample: whole frames (the oracle deals the reference's 200 chunks to `threads` workers) when a frame takes the
    # cores a few seconds at most; otherwise evenly spaced single rows, one worker thread per row at a time, as many rows as
    # make one pass take about four seconds (calibrated on the frame's middle row)
    from concurrent.futures import ThreadPoolExecutor
    shared_accum = np.zeros((height, width, 4), dtype=np.float32)

    def row(y: int) -> int:
        r = orc.render(scene.desc, cam, tex, accumulate=accumulate, n_threads=1, want_counters=True, rows=(y, y + 1), ray_dirs=dirs, accum=shared_accum)
        return sum(r["counters"][:5])

    t0 = time.perf_counter()
    row(height // 2)
    t_row = max(time.perf_counter() - t0, 1e-5)
    n_rows = int(max(threads, min(height, round(4.0 * threads / t_row))))
    starts = None if n_rows >= height else sorted({int((i + 0.5) * height / n_rows) for i in range(n_rows)})
    pool = ThreadPoolExecutor(max_workers=threads) if starts is not None else None

    def one_pass():
        """(rays, seconds) of one pass over the sample."""
        t0 = time.perf_counter()
        if starts is None:
            r = orc.render(scene.desc, cam, tex, accumulate=accumulate, n_threads=threads, want_counters=True, ray_dirs=dirs)
            rays = sum(r["counters"][:5])
        else:
            rays = sum(pool.map(row, starts))
        return rays, time.perf_counter() - t0

    one_pass()   # warm-up (page faults, thread start)
    # repeated for about twelve seconds, at least five times: the value is the median pass
    rates, rays_all, t_all = [], 0, 0.0
    while len(rates) < 5 or (t_all < 12.0 and len(rates) < 200):
        rays, dt = one_pass()
        rates.append(rays / dt / 1e6)
        rays_all += rays
        t_all += dt
    rates.sort()
    median = rates[len(rates) // 2] if len(rates) % 2 else 0.5 * (rates[len(rates) // 2 - 1] + rates[len(rates) // 2])
    what = f"full {width}x{height} frames" if starts is None else f"passes over {len(starts)} evenly spaced rows of the {width}x{height} frame"
    return {"value": round(median, 3), "unit": "Mrays/s", "cores": threads, "kind": "port", "min": round(rates[0], 3), "max": round(rates[-1], 3),
            "repeats": len(rates),
            "sample": f"{len(rates)} {what}: median {median:.4g}, min {rates[0]:.4g}, max {rates[-1]:.4g} Mrays/s; {rays_all} rays in {t_all:.2f} s; "
                      "CPU restatement of the reference (Rust toolchain unavailable)"}


if __name__ == "__main__":
    main()
