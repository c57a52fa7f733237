#!/usr/bin/env python3
"""Workgroups the tile classification marks as sky, per scene at 4K (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
for name in sys.argv[1:] or ("scene3", "scene1", "tex4", "scene4"):
    s = pkg.Scene(name)
    if not s.desc.ray_marching:
        s.desc.max_ray_bounces = 5
    ctx = pkg.Context(0); ctx.load(s, 3840, 2160)
    ctx.render_device(frame_index=1)
    ctx.stats()
    h = ctx.hierarchy_info()
    print(name, h["sky_workgroups"], "of", h["classified_workgroups"])
    ctx.close()
