#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04g; mkdir -p $OUT; rm -f $OUT/variants.txt
for lib in dmpfold2_amd/libdmpfold_hip.so tools/_bin/libdmp_vp_nomfma.so tools/_bin/libdmp_vp_nofinish.so tools/_bin/libdmp_vp_noload.so; do
  echo "== $lib" >> $OUT/variants.txt
  DMPFOLD_HIP_LIB=$PWD/$lib timeout 300 python tools/time_vgru_persist.py 8 300 2000 2>&1 | grep "vgru_persistent=1" >> $OUT/variants.txt
done
cat $OUT/variants.txt
python tools/single_trace.py run 300 2000 10 100 5 2>&1 | tail -3
