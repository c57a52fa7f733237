#!/bin/bash
# Runs on the GPU box: one default bench line per workload into gpurun_out/bench_<tag>/: tools/bench_all.sh <tag> [workloads...]
TAG=${1:-r03}; shift
WLS=${@:-scene3_4k scene1_4k tex4_4k scene4_4k scene2_4k spheres10k_4k spheres10k_8k scene3_800x600 scene3_4k_moving scene1_4k_animated}
mkdir -p gpurun_out/bench_$TAG
for wl in $WLS; do
  timeout -k 10 300 python3 bench.py --workload "$wl" > gpurun_out/bench_$TAG/$wl.json 2> gpurun_out/bench_$TAG/$wl.err || { echo "bench of $wl failed"; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_$TAG/$wl.json')); print('$wl', d['ms_per_step'], 'ms', d['value'], 'Mrays/s frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'cpu', d.get('cpu_baseline',{}).get('value'))"
done
