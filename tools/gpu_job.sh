#!/bin/bash
# scratch GPU job of the session (gpurun -- 'bash tools/gpu_job.sh'); every step under a timeout
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for lib in tools/_bin/libdmp_vp_nopipe.so tools/_bin/libdmp_vp_nocode.so tools/_bin/libdmp_vp_nohp.so dmpfold2_amd/libdmpfold_hip.so tools/_bin/libdmp_vp_pipe_nocode.so; do
  echo "== $lib"
  DMPFOLD_HIP_LIB=$PWD/$lib timeout 300 python tools/time_vgru_persist.py 8 300 2000 2>&1 | grep "vgru_persistent=1" | cut -c1-95,230-
done
