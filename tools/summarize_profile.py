#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into a small text summary (kept under profiles/)."""
import csv
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
    print("\n".join(out))


if __name__ == "__main__":
    main()
