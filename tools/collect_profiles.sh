#!/bin/bash
# Here, after tools/profile_all.sh has run on the GPU box: copies what is kept under profiles/ out of gpurun_out/
# (summaries, bench lines, the kernel statistics of the default workload) and merges the traffic entries.
# usage: tools/collect_profiles.sh <tag> [workloads...]
TAG=${1:-r03}; shift
WLS=${@:-scene3_4k scene1_4k tex4_4k scene4_4k scene2_4k spheres10k_4k spheres10k_8k scene3_800x600 scene3_4k_moving scene1_4k_animated}
cd "$(dirname "$0")/.." || exit 1
dirs=""
for wl in $WLS; do
  d=gpurun_out/prof_${TAG}_${wl}
  [ -f $d/summary.txt ] || { echo "no summary for $wl"; exit 1; }
  cp $d/summary.txt profiles/${TAG}_${wl}_summary.txt
  cp gpurun_out/bench_$TAG/$wl.json profiles/${TAG}_bench_${wl}.json
  dirs="$dirs $d"
done
for wl in scene3_4k spheres10k_8k; do
  d=gpurun_out/prof_${TAG}_${wl}
  [ -f $d/bench_under_trace.json ] && cp $d/bench_under_trace.json profiles/${TAG}_${wl}_bench_under_trace.json
done
f=$(ls gpurun_out/prof_${TAG}_scene3_4k/trace/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" profiles/${TAG}_scene3_4k_kernel_stats.csv
python3 tools/update_hbm_traffic.py merge $dirs
