#!/usr/bin/env python3
"""How much of a scene3 4K frame is the dispatch of its all-sky tile rows: the frame over all rows against the frame over the
rows from y0 down (rows above y0 hold nothing but sky for the y0 printed with 0 sky rays left above)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
s = pkg.Scene(sys.argv[1] if len(sys.argv) > 1 else "scene3"); s.desc.max_ray_bounces = 5
w, h = 3840, 2160
ctx = pkg.Context(0); ctx.load(s, w, h)
rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
def run(rows, n=300):
    for _ in range(20):
        ctx.render_device(frame_index=1, rows=rows, d_rgba8=rgba.data_ptr(), stream=stream, no_timing=True, flags=pkg.RT_RENDER_SKIP_ACCUM_STORE)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        ctx.render_device(frame_index=1, rows=rows, d_rgba8=rgba.data_ptr(), stream=stream, no_timing=True, flags=pkg.RT_RENDER_SKIP_ACCUM_STORE)
    b.record(); torch.cuda.synchronize()
    st = ctx.stats()
    return a.elapsed_time(b) / n, st["rays_total"], st["rays_sky_tiles"]
print("all rows", run(None))
for y0 in range(0, h, 120):
    ms, rays, sky = run((y0, h))
    print(f"rows {y0:4d}..{h}: {ms * 1e3:7.1f} us, rays {rays}, sky-tile rays {sky}")
# a fill of the rows above: torch's fill as a stand-in for a 16-byte-per-lane fill kernel
for y0 in (960, 1080, 1200, 1320):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    v = rgba.view(torch.int32)
    torch.cuda.synchronize(); a.record()
    for _ in range(300):
        v[:y0].fill_(0x00FF8040)
    b.record(); torch.cuda.synchronize()
    print(f"fill of rows 0..{y0}: {a.elapsed_time(b) / 300 * 1e3:.1f} us")
