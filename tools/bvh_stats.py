#!/usr/bin/env python3
"""Per-ray traversal statistics of the bounding-volume hierarchy on one frame: tools/bvh_stats.py scene width height"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s = pkg.Scene(name); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
ctx.render_device(frame_index=1, accumulate=True, seed=0x5EED, collect_events=True)
st = ctx.stats()
c = {k: st[k] for k in pkg.COUNTER_NAMES}
rays = st["rays_total"]
print("rays", rays, "per pixel", rays / (w * h))
for k, v in c.items():
    if v: print(f"  {k:24s} {v:14d}  {v / rays:8.2f} per ray")
for i in range(3):
    ctx.render_device(frame_index=2 + i, accumulate=True, seed=0x5EED)
    print("kernel ms", ctx.stats()["last_kernel_ms"])
