#!/usr/bin/env python3
"""Renders every scene on the GPU and on the CPU oracle at a small size and prints parity metrics."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 240)
    names = sys.argv[3:] or ["scene3", "scene1", "scene2", "scene4", "tex4"]
    ctx = pkg.Context(0)
    for name in names:
        s = pkg.Scene(name)
        ctx.load(s, w, h)
        acc = s.renderer_accumulation
        t0 = time.time()
        ctx.reset_accum()
        ctx.render_device(frame_index=1, accumulate=acc, collect_events=True)
        st = ctx.stats()
        gpu_acc = ctx.read_accum()
        gpu = ctx.render(frame_index=1, accumulate=acc, reset_accum=True)
        t1 = time.time()
        ref = orc.render(s.desc, s.camera(w, h), s.textures(), accumulate=acc, want_counters=True, n_threads=8)
        t2 = time.time()
        d = np.abs(gpu.astype(np.int32) - ref["rgba"].astype(np.int32))
        lin = np.abs(gpu_acc - ref["accum"])
        bitexact = int((gpu_acc.view(np.uint32) != ref["accum"].view(np.uint32)).sum())
        cref = dict(zip(pkg.COUNTER_NAMES, ref["counters"]))
        bad = {k: (st[k], cref[k]) for k in pkg.COUNTER_NAMES if st[k] != cref[k]}
        print(f"{name}: {w}x{h} gpu {t1 - t0:.2f}s cpu {t2 - t1:.2f}s  bytes within 1 LSB {100 * (d <= 1).mean():.4f}%  "
              f"bytes != {int((d > 0).sum())}  max lin err {np.nanmax(lin):.3e}  f32 words != {bitexact}  "
              f"rays {st['rays_total']}  kernel {st['last_kernel_ms']:.3f} ms  counter mismatches {bad}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
