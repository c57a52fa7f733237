#!/bin/bash
# Runs on the GPU box: tools/profile.sh for every bench workload, then one plain bench line per workload.
# usage: tools/profile_all.sh <tag> [workloads...]
TAG=${1:-r03}; shift
WLS=${@:-scene3_4k scene1_4k tex4_4k scene4_4k scene2_4k spheres10k_4k spheres10k_8k scene3_800x600 scene3_4k_moving scene1_4k_animated}
mkdir -p gpurun_out/bench_$TAG
for wl in $WLS; do
  bash tools/profile.sh "$TAG" "$wl" > gpurun_out/prof_${TAG}_${wl}.log 2>&1 || { echo "profile of $wl failed"; tail -3 gpurun_out/prof_${TAG}_${wl}.log; exit 1; }
  # the bench lines below report roofline.traffic from profiles/hbm_traffic.json when its entry was measured on this build
  python3 tools/update_hbm_traffic.py merge gpurun_out/prof_${TAG}_${wl} > /dev/null || { echo "no traffic entry for $wl"; exit 1; }
  echo "profiled $wl"
done
for wl in $WLS; do
  timeout -k 10 300 python3 bench.py --workload "$wl" > gpurun_out/bench_$TAG/$wl.json 2> gpurun_out/bench_$TAG/$wl.err || { echo "bench of $wl failed"; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_$TAG/$wl.json')); print('$wl', d['ms_per_step'], 'ms', d['value'], 'Mrays/s frac', d['roofline']['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'))"
done
