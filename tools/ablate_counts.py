#!/usr/bin/env python3
"""Scene-parameter ablations of one workload, one dispatch each, for per-dispatch PMC attribution:
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... -- python3 tools/ablate_counts.py scene3 3840 2160
Dispatch order: warm-up, full, no shadows, no lights, 1 bounce, 0 bounces, no objects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s = pkg.Scene(name); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
def frame(label):
    ctx.set_scene(s.desc)
    ctx.render_device(frame_index=1)
    st = ctx.stats()
    print(f"{label:12s} kernel {st['last_kernel_ms']:.4f} ms rays {st['rays_total']}")
frame("warm-up")
frame("full")
sc = s.desc.shadow_casting; s.desc.shadow_casting = 0; frame("no shadows"); s.desc.shadow_casting = sc
nl = s.desc.n_lights; s.desc.n_lights = 0; frame("no lights"); s.desc.n_lights = nl
s.desc.max_ray_bounces = 1; frame("1 bounce")
s.desc.max_ray_bounces = 0; frame("0 bounces")
s.desc.max_ray_bounces = 5; s.desc.n_objects = 0; frame("no objects")
