#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 300 python tools/debug_vgru_f32.py 40 > $OUT/dbg.txt 2>&1; grep -c "max|d|" $OUT/dbg.txt; grep "out vs h1" $OUT/dbg.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "vertical or gru_vertical" -s > $OUT/vgru_tests.log 2>&1; tail -5 $OUT/vgru_tests.log
VGRU_F32=1 timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $OUT/vgru_f32_time.txt 2>&1; tail -8 $OUT/vgru_f32_time.txt
timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $OUT/vgru_f16_time.txt 2>&1; tail -8 $OUT/vgru_f16_time.txt
