#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
for b in ubench_mfma_power_bf16 ubench_mfma_power_bf16_16 ubench_mfma_power_bf16_sameb; do echo "== $b"; timeout 120 tools/_bin/$b 2 2>&1 | grep -v "^small\|^zero"; done | tee $OUT/mfma_power_variants.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "minimiser_on and L500" > $OUT/l500.log 2>&1; tail -12 $OUT/l500.log | cut -c1-400
