#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
VGRU_F32=2 timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $OUT/vgru_x3.txt 2>&1; grep -E "oracle|=1: one chain|faults|bits" $OUT/vgru_x3.txt | cut -c1-600
