#!/bin/bash
# scratch GPU job of the session (gpurun -- 'bash tools/gpu_job.sh'); every step under a timeout
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "vertical or grouped or rider or persistent" 2>&1 | tail -3
for lib in tools/_bin/libdmp_prev.so dmpfold2_amd/libdmpfold_hip.so; do
  echo "== $lib"
  DMPFOLD_HIP_LIB=$PWD/$lib timeout 300 python tools/time_vgru_persist.py 8 300 2000 2>&1 | grep "vgru_persistent=1"
done
