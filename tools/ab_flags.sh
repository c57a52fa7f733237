#!/bin/bash
# A/B of bench.py options on the GPU box: tools/ab_flags.sh "<workloads>" "<flags A>" "<flags B>" ...
WLS=$1; shift
for r in 1 2; do
for wl in $WLS; do
  for f in "$@"; do
    timeout -k 10 150 python3 bench.py --workload $wl $f --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', '[$f]', 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'frac', d['roofline']['frac'])"
  done
done
done
