#!/bin/bash
# scratch GPU job of the session (gpurun -- 'bash tools/gpu_job.sh'); every step under a timeout
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 2400 bash tools/profile_r04.sh > $OUT/profile.log 2>&1
tail -12 $OUT/profile.log
