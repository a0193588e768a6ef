#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout; outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; OUT=$R/gpurun_out/job; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -q -x -s -k "backward or train" > $OUT/train_tests.log 2>&1; tail -4 $OUT/train_tests.log | cut -c 1-300
timeout 600 python tools/time_bwd.py 300 350 > $OUT/bwd_time.txt 2>&1; grep "^L=" $OUT/bwd_time.txt | cut -c 1-330
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/bwdprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bwdprof -o bwd -- python $R/tools/time_bwd.py 300 > /dev/null 2>&1
f=$(find /tmp/bwdprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bwd_kernel_stats.csv && head -6 "$f" | cut -c 1-200
