#!/usr/bin/env python3
"""ms of one rt_render_samples_device launch of the 8K 10 000-sphere frame for S = 1 ... 64 samples (what a rank renders when the
64 samples of a frame are split over 64 / S GPUs by sample index)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
w, h = 7680, 4320
s = pkg.Scene("spheres10k"); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for spp in (2, 4, 8, 16, 32, 64):
    times = []
    for rep in range(3):
        accum.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.render_samples_device(1 + 64 // spp, spp, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=st)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    print(f"{spp:2d} samples in one launch: {min(times):7.2f} ms  ({min(times) / spp:5.2f} ms per sample); 64 samples over {64 // spp} GPUs: x{194.0 / min(times):.2f} before the exchange")
# the other split: every rank renders all 64 samples of its interleaved 8-row bands (tile_row_stride = ranks); no accumulator exchange
for ranks in (2, 4, 8):
    worst = 0.0
    for r in range(ranks):
        times = []
        for rep in range(2):
            accum.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.render_samples_device(1, 64, rows=(8 * r, h), tile_row_stride=ranks, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=st)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        worst = max(worst, min(times))
    print(f"rows split over {ranks} GPUs, 64 samples each: slowest share {worst:7.2f} ms: x{194.0 / worst:.2f} before the all-gather of RGBA8")
