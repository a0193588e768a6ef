#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
for m in 0 2; do for L in 48 82 96 112 128 144 160 176 200 240; do for b in 1 2; do timeout 300 python tools/conv_only.py 40 $L $m $b 2>&1 | tail -1; done; done; done | tee $OUT/conv_bands.log
