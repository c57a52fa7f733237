#!/usr/bin/env python3
"""Keeps profiles/hbm_traffic.json tied to the library build it was measured on.

  collect <workload> <profile_dir>   (on the GPU box, at the end of tools/profile.sh) reads the FETCH_SIZE / WRITE_SIZE
                                     passes of that directory and writes <profile_dir>/hbm_entry.json with the bytes per
                                     render call and the sha256 of the librt_mi355x.so that ran
  merge <profile_dir> [...]          (here) copies those entries into profiles/hbm_traffic.json

bench.py reports `roofline.traffic` only when the stored sha256 equals the running library's.

Bytes per render call: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at
64 bytes, so the read side is doubled.  A render call may be several dispatches (the re-grouping path tracer launches an
order kernel in front of its render kernel): the sums over all timed render kernels are divided by the number of render calls, which is the
dispatch count of the kernel that runs once per call.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDER_KERNELS = ("rt_trace_kernel<false", "rt_march_kernel<false", "rt_wf_", "rt_refill_diffuse_kernel")   # timed variants; the counting variants are <true, ...>
ONCE_PER_CALL = ("rt_trace_kernel<false", "rt_march_kernel<false", "rt_wf_diffuse_kernel", "rt_wf_chain_kernel", "rt_refill_diffuse_kernel")


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def collect(workload, d):
    sums, calls = defaultdict(float), defaultdict(int)
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            c, k = r["Counter_Name"], r["Kernel_Name"]
            if c not in ("FETCH_SIZE", "WRITE_SIZE") or not any(t in k for t in RENDER_KERNELS):
                continue
            sums[c] += float(r["Counter_Value"])
            if any(t in k for t in ONCE_PER_CALL):
                calls[c] += 1
    if not calls.get("FETCH_SIZE") or not calls.get("WRITE_SIZE"):
        sys.exit(f"no FETCH_SIZE / WRITE_SIZE rows of a render kernel under {d}")
    rd = 2.0 * 1024.0 * sums["FETCH_SIZE"] / calls["FETCH_SIZE"]
    wr = 1024.0 * sums["WRITE_SIZE"] / calls["WRITE_SIZE"]
    lib = os.environ.get("RT_MI355X_LIB") or os.path.join(ROOT, "ray-tracing_amd", "librt_mi355x.so")
    entry = {"workload": workload, "bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
             "render_calls_measured": calls["FETCH_SIZE"], "lib_sha256": sha256(lib),
             "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py (tools/profile.sh); "
                       "read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; all kernels of one render call summed"}
    with open(os.path.join(d, "hbm_entry.json"), "w") as fh:
        json.dump(entry, fh, indent=1)
    print(json.dumps(entry))


def merge(dirs):
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        table = {}
    for d in dirs:
        e = json.load(open(os.path.join(d, "hbm_entry.json")))
        wl = e.pop("workload")
        e["profile_dir"] = os.path.basename(os.path.normpath(d))
        table[wl] = e
        print("merged", wl, e["bytes_per_launch"], e["lib_sha256"][:12])
    with open(path, "w") as fh:
        json.dump(table, fh, indent=1)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "collect":
        collect(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 3 and sys.argv[1] == "merge":
        merge(sys.argv[2:])
    else:
        sys.exit(__doc__)
