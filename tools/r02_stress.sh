#!/bin/bash
# GPU box: corruption stress of the scheduler with the current kernels (bit-exact against one engine),
# then the other BASELINE configurations.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02u
{
DBG_TRIALS=40 DBG_S=4 timeout 900 python tools/corrupt_repro.py
DBG_TRIALS=15 DBG_S=3 DBG_CONV_MODE=2 timeout 600 python tools/corrupt_repro.py
DBG_TRIALS=10 DBG_S=4 DBG_N=2000 DBG_IT=2 timeout 900 python tools/corrupt_repro.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02u/stress.txt
timeout 1500 python tools/bench_configs.py 2>&1 | grep "^{" | tee gpurun_out/r02u/configs.jsonl
