#!/bin/bash
# scratch GPU job of the session (gpurun -- 'bash tools/gpu_job.sh'); every step under a timeout
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
timeout 2400 bash tools/profile_r04.sh > $OUT/profile.log 2>&1
tail -30 $OUT/profile.log
