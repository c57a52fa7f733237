#!/bin/bash
# Memory-path counters of one workload's render kernel on the GPU box: tools/pmc_mem.sh <workload> [lib.so]
# (texture-address unit and L1 busy / stall cycles, L1 -> L2 requests and latencies), one rocprofv3 pass per counter group.
WL=$1; LIB=${2:-}
export TMPDIR=/tmp
[ -n "$LIB" ] && export RT_MI355X_LIB=$PWD/$LIB
tag=$(basename "${LIB:-default}" .so)
for pass in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  d=gpurun_out/pmcmem_${WL}_${tag}/$name; mkdir -p gpurun_out/pmcmem_${WL}_${tag}
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 bench.py --workload "$WL" --steps 4 --warmup 1 --no-cpu-baseline --consecutive-launches > /dev/null 2> $d.err || { echo "pass $name failed"; tail -2 $d.err; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcmem_${WL}_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if ("rt_trace_kernel<false" in k or "rt_march_kernel<false" in k or "rt_wf_diffuse" in k or "rt_wf_chain" in k or "rt_refill_diffuse" in k):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(agg.items()):
    print(f"  {c:40s} {sum(v)/len(v):18.0f}")
PY
