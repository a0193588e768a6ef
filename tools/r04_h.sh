#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward or grouped or rider or persistent" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -15 $OUT/tests.log
timeout 600 python -m pytest tests/test_gpu_headline.py -q -m gpu -x -k "full_mds_gain or scheduler or pipeline" > $OUT/tests2.log 2>&1
echo "tests2 rc=$?" >> $OUT/tests2.log
tail -8 $OUT/tests2.log
