#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 1700 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
bash tools/profile_r04.sh > $OUT/profile.log 2>&1
tail -40 $OUT/profile.log
