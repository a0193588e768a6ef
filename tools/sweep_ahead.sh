#!/bin/bash
# GPU box: look-ahead of the next group's vertical-GRU chain (enqueued by the helper thread) against none
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { label=$1; shift
  out=$(env "$@" python bench.py --no-cpu-baseline --no-exact-f32 --steps ${STEPS:-4} --warmup 1 2>/dev/null)
  python3 - "$label" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-28s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f  ok %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["verify"]["ok"]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-300:])
PY
}
run "lookahead 0" DMP_VGRU_LOOKAHEAD=0
for k in ${AHEAD:-1 3 8 16 32}; do run "lookahead $k" DMP_VGRU_LOOKAHEAD=$k; done
run "lookahead 0 (again)" DMP_VGRU_LOOKAHEAD=0
