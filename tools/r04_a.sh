#!/bin/bash
# round 4, first GPU call: new small-activation / hot-weights tests, parity budget
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -m gpu -k "small_activation or unscaled_pieces or f16_range or hot_target or second_weight or north_star_size_and_depth or headline_workload" -s > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
timeout 1500 python tools/parity_budget.py --cases w1x4,actsmall,actmixed,l300,fitns > $OUT/parity_budget.txt 2> $OUT/parity_budget.err
echo "budget rc=$?" >> $OUT/parity_budget.err
DMPFOLD_HIP_LIB=$PWD/tools/_bin/libdmp_accgates.so timeout 600 python tools/parity_budget.py --cases w1x4,l300,fitns --only none --modes 0,1 > $OUT/parity_budget_accgates.txt 2>> $OUT/parity_budget.err
tail -5 $OUT/tests.log
