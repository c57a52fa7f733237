#!/bin/bash
# A/B of one build under different RT_MI355X_REGION settings: tools/ab_env.sh "<workloads>" lib.so region1 region2 ...
WLS=$1; LIB=$2; shift 2
for reg in "$@"; do
  for wl in $WLS; do
    RT_MI355X_REGION=$reg RT_MI355X_LIB=$PWD/$LIB timeout -k 10 300 python3 bench.py --workload $wl --steps ${STEPS:-40} --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$reg', '$wl', 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'frac', d['roofline']['frac'])"
  done
done
