#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout; outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_headline.py -q -m gpu -k "missing_workgroup or launcher_over_rccl" > $OUT/few.log 2>&1; tail -4 $OUT/few.log | cut -c1-300
bash tools/profile_r06.sh > $OUT/profile.log 2>&1; tail -40 $OUT/profile.log | cut -c1-400
