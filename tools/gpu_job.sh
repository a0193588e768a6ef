#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; OUT=gpurun_out/job; mkdir -p $OUT
for rep in 1 2; do
  for v in 0 2; do
    DMP_GJ_DIAG=$v timeout 600 python bench.py --steps 8 --warmup 2 --legs f16x3 --no-cpu-baseline --no-files-leg > $OUT/gj_${v}_$rep.json 2> $OUT/gj_${v}_$rep.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/gj_${v}_$rep.json").read().strip().splitlines()[-1])
    print("gj_diag=$v rep $rep value %.3f chip_ms %.4f single %.1f ok %s digest_match %s" % (j["value"], j["roofline"]["chip_ms_per_launch"], j["single_target"]["ms"], j["verify"]["ok"], j["verify"].get("digest_match")))
except Exception as e:
    print("gj_diag=$v rep $rep FAILED", e); print(open("$OUT/gj_${v}_$rep.err").read()[-600:])
PY
  done
done
