#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_headline.py -q -x -k "missing_workgroup" > $OUT/drop.log 2>&1; tail -15 $OUT/drop.log | cut -c 1-250
