#!/bin/bash
# GPU box, round 3 session 4, call F: cluster tridiagonalisation (bitwise check, timing), L = 1344 reference golden
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "cluster_tridiag or largest_order or former_length" > $O/pytest_eig.txt 2>&1; grep -v "^$" $O/pytest_eig.txt | tail -15
timeout 300 python tools/gpu_diag.py --time 300 2000 > $O/diag_time.txt 2>&1; grep -E "eigh|coords_from|predict|gru_bidir" $O/diag_time.txt
timeout 300 python tools/single_trace.py run 300 2000 10 100 5 > $O/single.txt 2>&1; tail -3 $O/single.txt
