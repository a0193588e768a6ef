#!/bin/bash
# scratch GPU job of the session (gpurun -- 'bash tools/gpu_job.sh'); every step under a timeout
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/all.log 2>&1; tail -4 $OUT/all.log
for args in "--streams 2 --vgru-per-row" ""; do
echo "== $args"
timeout 600 python bench.py --steps 1 --warmup 1 --legs f16x3 --no-cpu-baseline $args 2>/dev/null | timeout 60 python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['verify']['ok'], d['single_target']['bitwise_equals_the_scheduler'], d['single_target']['cluster_tridiagonalisation_same_bits'], d['single_target']['ms'])"
done
