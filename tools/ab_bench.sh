#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by several per cent): alternating bench runs.
#   tools/ab_bench.sh <libA.so> <libB.so> [rounds] [extra bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-2}; shift 3
for i in $(seq $R); do
  for lib in "$A" "$B"; do
    DMPFOLD_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 6 --warmup 2 --cpu-baseline none --no-exact-f32 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$lib', 'value %.3f' % d['value'], 'chip_ms %.4f' % r['chip_ms_per_launch'], 'avg_launch %.4f' % r['avg_launch_ms'], 'in_flight %.2f' % r['launches_in_flight'], 'ok', d['verify']['ok'])"
  done
done
