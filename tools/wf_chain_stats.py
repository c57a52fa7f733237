#!/usr/bin/env python3
"""Lane occupancy per phase of the path tracer with sample chains (library built with -DRT_WF_STATS):
RT_MI355X_LIB=build_variants/x.so python3 tools/wf_chain_stats.py [width height [samples]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 16
s = pkg.Scene("spheres10k"); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
os.environ["RT_WF_STATS_PRINT"] = "1"
ctx.render_samples_device(1, samples, seed=0x5EED)
st = ctx.stats()
print("rays", st["rays_total"], "of them camera rays not traced", st["rays_reused_camera"])
