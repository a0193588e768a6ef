#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout; outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 2700 python -m pytest tests -x -q -m gpu > $OUT/all.log 2>&1; tail -5 $OUT/all.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
