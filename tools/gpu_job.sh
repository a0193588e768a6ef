#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout; outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/all.log 2>&1; tail -5 $OUT/all.log
bash tools/profile_r05.sh > $OUT/profile_r05.log 2>&1; tail -12 $OUT/profile_r05.log | cut -c 1-300
