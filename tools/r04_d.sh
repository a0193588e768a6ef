#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04d; mkdir -p $OUT; rm -f $OUT/vgru_variants.txt
timeout 300 python tools/time_vgru_persist.py 8 300 2000 2>&1 | grep -v amdgpu.ids >> $OUT/vgru_variants.txt
cat $OUT/vgru_variants.txt
