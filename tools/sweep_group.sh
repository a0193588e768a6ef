#!/bin/bash
# GPU box: scheduler policy sweep (same box, back to back): streams x group size x patience
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { # label, env..., -- bench args
  label=$1; shift
  out=$(env "$@" python $R/bench.py --no-cpu-baseline --no-exact-f32 --steps ${STEPS:-3} --warmup 1 ${BARGS} 2>/dev/null)
  python3 - "$label" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-44s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f  ok %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["verify"]["ok"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
BARGS="--streams 4" run "S=4 G=4 patience 40" DMP_VGRU_GROUP=4
BARGS="--streams 4" run "S=4 G=1" DMP_VGRU_GROUP=1
BARGS="--streams 4" run "S=4 G=4 patience 16" DMP_VGRU_GROUP=4 DMP_GROUP_PATIENCE=16
BARGS="--streams 4" run "S=4 G=4 patience 100" DMP_VGRU_GROUP=4 DMP_GROUP_PATIENCE=100
BARGS="--streams 4" run "S=4 G=2" DMP_VGRU_GROUP=2
BARGS="--streams 6 --batch 12" run "S=6 G=3 (batch 12)" DMP_VGRU_GROUP=3 GPU_MAX_HW_QUEUES=8
BARGS="--streams 8 --batch 8" run "S=8 G=4 (batch 8)" DMP_VGRU_GROUP=4 GPU_MAX_HW_QUEUES=12
BARGS="--streams 5 --batch 10" run "S=5 G=4 (batch 10)" DMP_VGRU_GROUP=4
BARGS="--streams 4" run "S=4 G=4 patience 40 (again)" DMP_VGRU_GROUP=4
