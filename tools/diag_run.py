#!/usr/bin/env python3
"""Prints the per-section wave-cycle breakdown of one frame (needs a library built with -DRT_DIAG via RT_MI355X_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s = pkg.Scene(name); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
for i in range(3):
    ctx.render_device(frame_index=1)
    ctx.stats()
os.environ["RT_DIAG_PRINT"] = "1"
ctx.render_device(frame_index=1)
print(ctx.stats()["last_kernel_ms"], "ms")
