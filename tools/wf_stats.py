#!/usr/bin/env python3
"""Lane occupancy per phase of the re-grouping path tracer (library built with -DRT_WF_STATS):
RT_MI355X_LIB=build_variants/x.so RT_WF_STATS_PRINT=1 python3 tools/wf_stats.py [width height]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
s = pkg.Scene("spheres10k"); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
ctx.render_device(frame_index=1, accumulate=True, seed=0x5EED)
print("rays", ctx.stats()["rays_total"])
