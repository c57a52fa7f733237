#!/usr/bin/env python3
"""Frame time of the ray-marched scene at 4K: default view, all sky, all ground (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
s = pkg.Scene("scene4")
w, h = 3840, 2160
ctx = pkg.Context(0); ctx.load(s, w, h)
rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
views = {"default": None, "sky": ((0.0, 2.0, 4.0), (0.0, 5.0, -1.0)), "ground": ((0.0, 2.0, 4.0), (0.0, -5.0, -1.0))}
for name, v in views.items():
    if v is not None:
        ctx.set_camera(pkg.make_camera(v[0], v[1], w, h))
    for _ in range(100):
        ctx.render_device(frame_index=1, d_rgba8=rgba.data_ptr(), flags=pkg.RT_RENDER_SKIP_ACCUM_STORE, no_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        ctx.render_device(frame_index=1, d_rgba8=rgba.data_ptr(), flags=pkg.RT_RENDER_SKIP_ACCUM_STORE, no_timing=True)
    torch.cuda.synchronize()
    print(name, f"{(time.perf_counter() - t0) / n * 1e3:.4f} ms")
