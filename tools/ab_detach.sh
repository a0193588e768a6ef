#!/bin/bash
# GPU box: detached vertical-GRU chain (helper thread + own stream) against the leader's-unit form, alternating runs
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { label=$1; shift
  out=$(env "$@" python $R/bench.py --no-cpu-baseline --no-exact-f32 --steps ${STEPS:-3} --warmup 1 2>/dev/null)
  python3 - "$label" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-28s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f  ok %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["verify"]["ok"]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-300:])
PY
}
for i in 1 2; do
  run "detach=1" DMP_VGRU_DETACH=1
  run "detach=0" DMP_VGRU_DETACH=0
done
run "detach=1 queues=12" DMP_VGRU_DETACH=1 GPU_MAX_HW_QUEUES=12
