#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4j; mkdir -p $O; cd $R
bash tools/r03s4_h.sh > $O/tc_prof_run.txt 2>&1; tail -6 $O/tc_prof_run.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cluster_tridiag or largest_order" > $O/pytest_eig.txt 2>&1; tail -3 $O/pytest_eig.txt
timeout 300 python tools/single_trace.py run 300 2000 10 100 5 > $O/single.txt 2>&1; tail -2 $O/single.txt
STEPS=4 timeout 900 bash tools/ab_env.sh DMP_TRIDIAG_CLUSTER 0 1 2 > $O/ab_cluster.txt 2>&1; cat $O/ab_cluster.txt
