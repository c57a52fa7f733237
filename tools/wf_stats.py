#!/usr/bin/env python3
"""Lane occupancy per phase and the launch's tail of the re-grouping path tracer (library built with -DRT_WF_STATS):
RT_MI355X_LIB=build_variants/x.so python3 tools/wf_stats.py [width height [frames]]   (prints for the last frame)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
s = pkg.Scene("spheres10k"); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
for f in range(1, frames + 1):
    if f == frames:
        os.environ["RT_WF_STATS_PRINT"] = "1"
    ctx.render_device(frame_index=f, accumulate=True, seed=0x5EED)
    ctx.stats()
print("rays", ctx.stats()["rays_total"])
