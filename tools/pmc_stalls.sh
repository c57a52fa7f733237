#!/bin/bash
# Instruction-cache, scalar-cache and latency counters of one workload (separate --pmc runs): tools/pmc_stalls.sh <workload>
WL=${1:-scene3_4k}
OUT=gpurun_out/stalls_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
            "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline --consecutive-launches > /dev/null 2> "$OUT/pmc_$name.err" || { echo "pmc pass $name failed"; tail -3 "$OUT/pmc_$name.err"; exit 1; }
done
python3 tools/summarize_profile.py "$OUT" | grep -v "true," | grep -A12 "counters"
