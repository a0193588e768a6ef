#!/bin/bash
# GPU box: bench A/B on one box, alternating - the tree at the start of this session (tools/_old, commit fc38834) against HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4i; mkdir -p $O; cd $R
run() { ( cd $1 && out=$(env $3 python bench.py --no-cpu-baseline --no-exact-f32 --steps ${STEPS:-4} --warmup 1 2>/dev/null); python3 - "$2" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-34s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f  ok %s digest %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["verify"]["ok"], j["verify"].get("digest", "")[:12]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-300:])
PY
) }
for i in 1 2; do
  run $R/tools/_old "old (fc38834)" "X=1"
  run $R "new, riders 4 (default)" "DMP_VGRU_RIDERS=4"
  run $R "new, riders 0" "DMP_VGRU_RIDERS=0"
done > $O/ab.txt 2>&1
cat $O/ab.txt
