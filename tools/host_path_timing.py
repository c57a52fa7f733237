#!/usr/bin/env python3
"""Wall time of the host-buffer entry point rt_render (kernel + device-to-host copy): tools/host_path_timing.py scene w h"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s = pkg.Scene(name); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
import numpy as np
img = np.zeros((h, w, 4), dtype=np.uint8)
for i in range(3):
    ctx.render(frame_index=1, reset_accum=True, out=img)
n = 20
def timed(label):
    t0 = time.perf_counter()
    for i in range(n):
        ctx.render(frame_index=1, reset_accum=False, out=img)
    dt = (time.perf_counter() - t0) / n
    st = ctx.stats()
    print(label, f"rt_render wall {dt * 1e3:.3f} ms/frame, kernel {st['last_kernel_ms']:.3f} ms, copy {st['last_copy_ms']:.3f} ms, {w * h * 4 / dt / 1e9:.2f} GB/s frame delivery")
timed("pageable:")
ref = img.copy()
img = ctx.alloc_host_frame()
timed("pinned:  ")
assert (img == ref).all()
ctx.free_host_frame(img)
sys.exit(0)
print(f"rt_render wall {dt * 1e3:.3f} ms/frame, kernel {st['last_kernel_ms']:.3f} ms, copy {st['last_copy_ms']:.3f} ms, {w * h * 4 / dt / 1e9:.2f} GB/s frame delivery")
