#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_headline.py -q -x -k "missing_workgroup or chains_time_out" > $OUT/drop.log 2>&1; tail -25 $OUT/drop.log | cut -c 1-250
