#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into a small text summary (kept under profiles/)."""
import csv
import json
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    out = []
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        out.append(f"== kernel stats ({os.path.relpath(f, d)})")
        out.append(open(f).read().strip())
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        by = defaultdict(list)
        for r in rows:
            by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
        out.append("== per-kernel dispatch durations (ns): name, n, mean, min, max, VGPR, SGPR, LDS, scratch, grid, wg")
        for k, v in by.items():
            ds = [x[0] for x in v]
            r = v[0][1]
            out.append(f"{k[:90]}, {len(ds)}, {sum(ds) / len(ds):.0f}, {min(ds)}, {max(ds)}, {r.get('VGPR_Count')}, {r.get('SGPR_Count')}, "
                       f"{r.get('LDS_Block_Size')}, {r.get('Scratch_Size')}, {r.get('Grid_Size')}, {r.get('Workgroup_Size')}")
        # the timed steps of bench.py are the last `steps` dispatches of the render kernel before the counting variant's
        # single dispatch; the ones before them are the clock pre-warm and the warm-up steps
        after = 1
        try:
            line = json.load(open(os.path.join(d, "bench_under_trace.json")))
            steps = line["steps"] * (1 if "rt_wf_chain" in line["roofline"]["kernel"] else line["config"]["samples_per_step"])
            after = int(line["config"].get("render_launches_after_timed_steps", 1))
        except (OSError, ValueError, KeyError):
            steps = None
        if steps:
            for k, v in by.items():
                if "rt_" in k and ("<false" in k or "rt_wf_chain" in k or "rt_wf_diffuse" in k or "rt_refill" in k) and len(v) >= steps + after:
                    v.sort(key=lambda x: int(x[1]["Start_Timestamp"]))
                    timed = [x[0] for x in v[-(steps + after):-after]]   # behind them: bench.py's untimed passes (ray counts, store variant)
                    out.append(f"== timed steps only: {k[:90]}: {len(timed)} dispatches, {after} more of the kernel behind them, mean {sum(timed) / len(timed):.0f} ns, "
                               f"min {min(timed)}, max {max(timed)}")
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append(f"== counters ({os.path.relpath(f, d)}): mean per dispatch")
        for k, cs in agg.items():
            if "rt_" not in k:
                continue
            out.append(f"  {k[:90]} (dispatches {len(next(iter(cs.values())))})")
            for c, v in sorted(cs.items()):
                out.append(f"    {c} = {sum(v) / len(v):.1f}")
    # HBM traffic per dispatch of the render kernels: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
    # 128-byte read requests at 64 bytes, so the read side is doubled.
    traffic = defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and "rt_" in r["Kernel_Name"]:
                traffic[r["Kernel_Name"]].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, cs in traffic.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            rd = 2.0 * 1024.0 * sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
            wr = 1024.0 * sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            out.append(f"== HBM traffic per dispatch, {k[:90]}: read {rd:.0f} B (2 x FETCH_SIZE KiB), write {wr:.0f} B, total {rd + wr:.0f} B")
    print("\n".join(out))


if __name__ == "__main__":
    main()
