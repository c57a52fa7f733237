#!/bin/bash
# HBM-side bytes per render launch of one workload for one library build (GPU box): tools/pmc_traffic.sh <workload> [lib.so]
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (KiB; the read side doubled on gfx950).
WL=$1; LIB=${2:-}
export TMPDIR=/tmp
[ -n "$LIB" ] && export RT_MI355X_LIB=$PWD/$LIB
tag=$(basename "${LIB:-default}" .so)
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmctr_${WL}_${tag}/$c; mkdir -p $d
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline ${PMC_BENCH_ARGS:-} > /dev/null 2> $d.err || { echo "pass $c failed"; tail -2 $d.err; exit 1; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmctr_${WL}_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if ("rt_trace_kernel<false" in k or "rt_march_kernel<false" in k or "rt_wf_" in k or "rt_refill_diffuse" in k):
            agg[(k.split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k,c),v in sorted(agg.items()):
    mb = (2.0 if c == "FETCH_SIZE" else 1.0) * 1024.0 * sum(v) / len(v) / 1e6
    print(f"${tag} ${WL} {k:60s} {c:10s} {mb:12.2f} MB per launch ({len(v)} launches)")
PY
