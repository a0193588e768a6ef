#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; OUT=gpurun_out/job; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "blocked_diagonal" > $OUT/inv_tests.log 2>&1; grep "D=\|passed\|failed" $OUT/inv_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_headline.py -q -k "config1_L200 or baseline_config_sizes or second_weight_set" > $OUT/three.log 2>&1; tail -6 $OUT/three.log; grep -n "^E   " $OUT/three.log | head
timeout 300 python tools/time_inverse.py 300 > $OUT/inverse.txt 2>&1; grep "pairs \|chain128" $OUT/inverse.txt
