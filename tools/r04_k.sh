#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04k; mkdir -p $OUT; rm -f $OUT/ab.txt
for rep in 1 2; do
for lib in tools/_bin/libdmp_prev.so dmpfold2_amd/libdmpfold_hip.so; do
  echo "== $lib" >> $OUT/ab.txt
  DMPFOLD_HIP_LIB=$PWD/$lib timeout 300 python tools/single_trace.py run 300 2000 10 100 6 2>&1 | grep "prediction" | tail -3 >> $OUT/ab.txt
done
done
cat $OUT/ab.txt
