#!/usr/bin/env python3
"""Generates tests/golden/oracle_frames.npz: 64x48 frames of the five test scenes from the CPU oracle.

The reference holds no golden vectors and cannot be run in this build, so these fixtures come from this
repository's own restatement (oracle/liborc.so, deterministic libm) and pin it against regressions; they are
also what the GPU tests compare with when they check the committed data rather than a live oracle run.
Frames: scene1, scene3, scene4, tex4 at frame_index 1 (deterministic), scene2 accumulated over frames 1..3 with
seed 0x5EED.  Textures are decoded from tests/golden/textures with PIL.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

W, H = 64, 48


def main():
    pkg, orc = g.load_package(), g.load_oracle()
    out = {}
    for name in ["scene1", "scene3", "scene4", "tex4"]:
        s = pkg.Scene(name)
        r = orc.render(s.desc, s.camera(W, H), s.textures(), want_counters=True)
        out[f"{name}_rgba"] = r["rgba"]
        out[f"{name}_accum"] = r["accum"]
        out[f"{name}_counters"] = np.array(r["counters"][: pkg.RT_CNT_COUNT], dtype=np.uint64)
    s = pkg.Scene("scene2")
    acc = np.zeros((H, W, 4), dtype=np.float32)
    for f in (1, 2, 3):
        r = orc.render(s.desc, s.camera(W, H), (), frame_index=f, accumulate=True, seed=0x5EED, accum=acc)
    out["scene2_rgba"] = r["rgba"]
    out["scene2_accum"] = acc
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_frames.npz"), **out)
    print("wrote tests/golden/oracle_frames.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
