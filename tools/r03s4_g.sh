#!/bin/bash
# GPU box: kernel durations of the eigensolver at n = 300 (rocprofv3 kernel trace of the stage timing tool)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/eprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eprof -o e -- python $R/tools/gpu_diag.py --time 300 2000 > $O/run.txt 2>&1
f=$(find /tmp/eprof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
grep -E "tridiag|tri_eig|backtransform|eig_load|seq_gru|refine_cluster|gemm_kernel<false, true, true>" $O/kernel_stats.csv | cut -c1-200
