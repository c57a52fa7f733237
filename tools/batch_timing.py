#!/usr/bin/env python3
"""ms per 64-spp 8K frame of the 10 000-sphere scene: 64 rt_render_device calls against one rt_render_samples_device call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
s = pkg.Scene("spheres10k"); s.desc.max_ray_bounces = 5
ctx = pkg.Context(0); ctx.load(s, w, h)
rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def frame(batched):
    accum.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    if batched:
        ctx.render_samples_device(1, spp, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=st)
    else:
        for k in range(1, spp + 1):
            ctx.render_device(frame_index=k, accumulate=True, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(), stream=st, no_timing=True)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 2)
frame(False); one = [frame(False) for _ in range(2)]; ref = accum.clone()
frame(True); two = [frame(True) for _ in range(3)]
print("consecutive calls", one, "ms; one batch call", two, "ms; identical:", bool(torch.equal(ref, accum)))
