#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout; outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 2700 python -m pytest tests -q -m gpu > $OUT/all.log 2>&1; tail -15 $OUT/all.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-files-leg --cpu-baseline none > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/job/bench.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','value_f32','value_split_f16')}, d['verify']['ok'], d['verify'])
PY
