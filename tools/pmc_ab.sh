#!/bin/bash
# PMC comparison of kernel builds on the GPU box: tools/pmc_ab.sh <workload> lib1.so lib2.so ...
WL=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
              "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
              "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" \
              "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU_CVT SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT"; do
    name=$(echo "$pass" | cut -d' ' -f1)
    d=gpurun_out/pmcab_${WL}_${tag}/$name; mkdir -p gpurun_out/pmcab_${WL}_${tag}
    RT_MI355X_LIB=$PWD/$lib timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 bench.py --workload "$WL" --steps 4 --warmup 1 --no-cpu-baseline --consecutive-launches > /dev/null 2> $d.err || { echo "pass $name failed for $tag"; tail -2 $d.err; }
  done
  echo "== $tag"
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcab_${WL}_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if ("rt_trace_kernel<false" in k or "rt_march_kernel<false" in k):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(agg.items()):
    print(f"  {c:28s} {sum(v)/len(v):16.0f}")
PY
done
