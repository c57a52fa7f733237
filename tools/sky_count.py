import os, sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
pkg = g.load_package()
for name in ("scene3", "scene1", "tex4", "scene2"):
    s = pkg.Scene(name); s.desc.max_ray_bounces = 5
    ctx = pkg.Context(0); ctx.load(s, 3840, 2160)
    ctx.render_device(frame_index=1, flags=pkg.RT_RENDER_ONE_LANE_PER_PIXEL)
    ctx.stats()
    print(name, ctx.hierarchy_info())
    ctx.close()
