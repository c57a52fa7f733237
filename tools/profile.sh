#!/bin/bash
# Runs on the GPU box: kernel-trace statistics and (separately) PMC passes of the default bench workload.
# usage: tools/profile.sh <tag> [workload]
set -u
TAG=${1:-r01}
WL=${2:-scene3_4k}
OUT=gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --workload "$WL" --no-cpu-baseline > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || exit 1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM" \
            "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
            "FETCH_SIZE" \
            "WRITE_SIZE" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$name.err" || { echo "pmc pass $name failed"; tail -3 "$OUT/pmc_$name.err"; exit 1; }
done
python3 tools/update_hbm_traffic.py collect "$WL" "$OUT" > /dev/null || exit 1
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.txt"
echo done
