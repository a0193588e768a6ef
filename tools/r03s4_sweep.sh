#!/bin/bash
# GPU box: scheduler knobs with riders, alternating bench runs on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { out=$(env $1 python bench.py --no-cpu-baseline --no-exact-f32 --steps 4 --warmup 1 2>/dev/null)
  python3 - "$1" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-36s %.3f structures/s  chip_ms/launch %.4f  ok %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["verify"]["ok"]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-200:])
PY
}
for i in 1 2; do
  run DMP_GROUP_PATIENCE=64
  run DMP_GROUP_PATIENCE=96
  run DMP_GROUP_PATIENCE=64
  run DMP_GROUP_PATIENCE=176
  run DMP_GROUP_PATIENCE=40
done
