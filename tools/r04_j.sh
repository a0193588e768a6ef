#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -25 $OUT/tests.log
