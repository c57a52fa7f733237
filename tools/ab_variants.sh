#!/bin/bash
# A/B of kernel builds on the GPU box: tools/ab_variants.sh "<workloads>" lib1.so lib2.so ...
# Stops at the first run that fails (a faulting build must not be followed by more GPU work).
WLS=$1; shift
set -o pipefail
for r in 1 2; do
for lib in "$@"; do
  for wl in $WLS; do
    RT_MI355X_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $wl ${ABSTEPS:---steps 200 --warmup 20} --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', '$wl', 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'frac', d['roofline']['frac'])" || { echo "FAILED: $lib $wl -- stopping"; exit 1; }
  done
done
done
