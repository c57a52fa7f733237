#!/bin/bash
# Core SQ / memory-path counters of the render kernel for one workload, few steps: tools/pmc_wf.sh <workload> <steps> [lib.so]
WL=$1; STEPS=$2; LIB=${3:-}
export TMPDIR=/tmp
[ -n "$LIB" ] && export RT_MI355X_LIB=$PWD/$LIB
tag=$(basename "${LIB:-default}" .so)
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" \
            "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_WRITE_sum TD_TD_BUSY_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  d=gpurun_out/pmcwf_${WL}_${tag}/$name; mkdir -p gpurun_out/pmcwf_${WL}_${tag}
  timeout -s KILL 170 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 bench.py --workload "$WL" --steps $STEPS --warmup 0 --no-cpu-baseline ${PMC_BENCH_ARGS---consecutive-launches} > /dev/null 2> $d.err || { echo "pass $name failed"; tail -2 $d.err; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcwf_${WL}_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if ("rt_trace_kernel<false" in k or "rt_march_kernel<false" in k or "rt_wf_diffuse" in k or "rt_wf_chain" in k or "rt_refill_diffuse" in k):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
g=lambda c: sum(agg[c])/len(agg[c]) if agg.get(c) else float('nan')
for c,v in sorted(agg.items()):
    print(f"  {c:40s} {sum(v)/len(v):18.0f}")
cyc=g("GRBM_GUI_ACTIVE")/8
print(f"  kernel cycles (per XCD)        {cyc:14.0f}")
print(f"  lane utilisation               {g('SQ_THREAD_CYCLES_VALU')/(64*g('SQ_ACTIVE_INST_VALU')):.3f}")
print(f"  VALU busy                      {4*g('SQ_ACTIVE_INST_VALU')/(1024*cyc):.3f}")
print(f"  TA busy                        {g('TA_TA_BUSY_sum')/(256*cyc):.3f}")
print(f"  TCP clocked                    {g('TCP_GATE_EN1_sum')/(256*cyc):.3f}")
print(f"  TCP line accesses per cycle    {g('TCP_TOTAL_CACHE_ACCESSES_sum')/(256*cyc):.3f}")
print(f"  L1 miss rate                   {g('TCP_TCC_READ_REQ_sum')/g('TCP_TOTAL_CACHE_ACCESSES_sum'):.3f}")
print(f"  L2 read latency (cycles)       {g('TCP_TCC_READ_REQ_LATENCY_sum')/g('TCP_TCC_READ_REQ_sum'):.0f}")
print(f"  L2 hit rate                    {g('TCC_HIT_sum')/(g('TCC_HIT_sum')+g('TCC_MISS_sum')):.3f}")
PY
