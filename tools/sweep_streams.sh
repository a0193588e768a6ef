#!/bin/bash
# GPU box: engines per GPU x hardware queues (two groups of four whose chains may run side by side)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { label=$1; shift; args=$1; shift
  out=$(env "$@" python bench.py --no-cpu-baseline --no-exact-f32 --steps 3 --warmup 1 $args 2>/dev/null)
  python3 - "$label" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-40s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f  ok %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["verify"]["ok"]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-300:])
PY
}
run "S=4 batch 8" "--streams 4" GPU_MAX_HW_QUEUES=8
run "S=8 batch 16 queues 12" "--streams 8 --batch 16" GPU_MAX_HW_QUEUES=12
run "S=8 batch 16 queues 16" "--streams 8 --batch 16" GPU_MAX_HW_QUEUES=16
run "S=6 batch 12 queues 8" "--streams 6 --batch 12" GPU_MAX_HW_QUEUES=8
run "S=8 batch 16 queues 12 detach" "--streams 8 --batch 16" GPU_MAX_HW_QUEUES=12 DMP_VGRU_DETACH=1
run "S=4 batch 8 (again)" "--streams 4" GPU_MAX_HW_QUEUES=8
