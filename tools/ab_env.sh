#!/bin/bash
# GPU box: A/B of one environment switch of the scheduler, alternating bench runs on one box
#   tools/ab_env.sh VAR valueA valueB [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$1; A=$2; B=$3; N=${4:-2}
run() { out=$(env $V=$1 python bench.py --no-cpu-baseline --no-exact-f32 --steps ${STEPS:-5} --warmup 1 2>/dev/null)
  python3 - "$V=$1" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-28s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f  ok %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["verify"]["ok"]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-300:])
PY
}
for i in $(seq $N); do run $A; run $B; done
