#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02c
for i in 1 2; do ./tools/_bin/ubench_conv_rr 24 300; done > gpurun_out/r02c/conv_rr.txt 2>&1
UB_DATA=zero ./tools/_bin/ubench_conv_rr 24 300 | grep "ms avg" >> gpurun_out/r02c/conv_rr.txt 2>&1
UB_DATA=f16 ./tools/_bin/ubench_conv_rr 24 300 | grep "ms avg" >> gpurun_out/r02c/conv_rr.txt 2>&1
cat gpurun_out/r02c/conv_rr.txt
timeout 600 python -m pytest tests/test_gpu_headline.py -m gpu -q -k "fault or hot or isolates or residue" 2>&1 | tail -15 > gpurun_out/r02c/pytest.log
cat gpurun_out/r02c/pytest.log
