#!/usr/bin/env python3
"""Time of one 8K sample of the 10 000-sphere scene over different row ranges (tail / balance experiments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
w, h = 7680, 4320
s = pkg.Scene("spheres10k"); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda"); acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream()
def run(rows, n=8):
    for i in range(2): ctx.render_device(frame_index=1 + i, accumulate=True, rows=rows, d_rgba8=rgba.data_ptr(), d_accum=acc.data_ptr(), stream=st.cuda_stream, no_timing=True)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(st)
    for i in range(n): ctx.render_device(frame_index=3 + i, accumulate=True, rows=rows, d_rgba8=rgba.data_ptr(), d_accum=acc.data_ptr(), stream=st.cuda_stream, no_timing=True)
    e1.record(st); torch.cuda.synchronize()
    ctx.render_device(frame_index=1, accumulate=True, rows=rows, d_rgba8=rgba.data_ptr(), d_accum=acc.data_ptr(), stream=st.cuda_stream)
    return e0.elapsed_time(e1) / n, ctx.stats()["rays_total"]
for rows in [(0, 4320), (0, 2160), (2160, 4320), (2160, 3240), (3240, 4320), (3240, 3784), (3784, 4320)]:
    ms, rays = run(rows)
    print(f"rows {rows}: {ms:.3f} ms, {rays/1e6:.1f} Mrays, {rays/ms/1e6:.2f} Grays/s")
