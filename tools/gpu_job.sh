#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
DMP_GJ_DIAG_BLOCKED=1 timeout 2400 python -m pytest tests -q -m gpu > $OUT/all_blocked.log 2>&1; tail -30 $OUT/all_blocked.log | cut -c1-300
