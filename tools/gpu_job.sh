#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 8 --warmup 2 --legs f16x3 --no-cpu-baseline --no-files-leg > $OUT/sw_$tag.json 2> $OUT/sw_$tag.err
python - <<PY
import json
try:
    j = json.loads(open("$OUT/sw_$tag.json").read().strip().splitlines()[-1])
    print("$tag value %.3f chip_ms %.4f ok %s" % (j["value"], j["roofline"]["chip_ms_per_launch"], j["verify"]["ok"]))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run base1 X=1
run pat20 DMP_GROUP_PATIENCE=20
run pat80 DMP_GROUP_PATIENCE=80
run base2 X=1
run stag4 DMP_TAIL_STAGGER=4
run stag12 DMP_TAIL_STAGGER=12
run base3 X=1
run depth3 DMP_LANE_DEPTH=3
run base4 X=1
