#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout; outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -s > $OUT/train_tests.log 2>&1; tail -30 $OUT/train_tests.log | cut -c 1-400
timeout 600 python -m pytest tests/test_gpu_headline.py -q -x -k "env_selects" > $OUT/env_test.log 2>&1; tail -3 $OUT/env_test.log
