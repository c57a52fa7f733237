#!/usr/bin/env python3
"""Experiment: the 8K / 10 000-sphere / 64-spp frame rendered as two interleaved half-frames by two contexts on two streams,
so that the tail of one launch is filled by the head of the other.  Prints ms per 64-spp frame for one and for two streams
and checks that both give the same accumulator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 2
s = pkg.Scene("spheres10k"); s.desc.max_ray_bounces = 5
ctxs = [pkg.Context(0) for _ in range(parts)]
for c in ctxs:
    c.load(s, w, h)
streams = [torch.cuda.Stream() for _ in range(parts)]
rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")

def frame(n_parts):
    accum.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, spp + 1):
        for p in range(n_parts):
            ctxs[p].render_device(frame_index=k, accumulate=True, seed=0x5EED, rows=(8 * p, h) if n_parts > 1 else None,
                                  tile_row_stride=n_parts if n_parts > 1 else 0, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr(),
                                  stream=streams[p].cuda_stream, no_timing=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

frame(1)
one = [frame(1) for _ in range(2)]
ref = accum.clone(); ref_rgba = rgba.clone()
frame(parts)
two = [frame(parts) for _ in range(2)]
print("one stream", one, "ms;", parts, "streams", two, "ms; identical accumulators:", bool(torch.equal(ref, accum)), bool(torch.equal(ref_rgba, rgba)))
