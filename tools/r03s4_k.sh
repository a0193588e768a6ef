#!/bin/bash
# GPU box: the headline fixture (L=300, N=2000, n=10, m=100 against the reference) by gate functions
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4k; mkdir -p $O; cd $R
for lib in dmpfold2_amd/libdmpfold_hip.so tools/_bin/libvg_acc.so tools/_bin/libseq_libm.so; do
  echo "== $lib"
  DMPFOLD_HIP_LIB=$R/$lib timeout 600 python -m pytest tests/test_gpu_headline.py -m gpu -q -s -k "headline_workload_with_minimiser" 2>&1 | grep -v "^$" | tail -9
  DMPFOLD_HIP_LIB=$R/$lib python tools/time_seq_gru.py 300 | tail -2
  DMPFOLD_HIP_LIB=$R/$lib python tools/time_vgru_group.py 4 300 2000 2>&1 | grep "ONE chain"
done > $O/gates2.txt 2>&1
cat $O/gates2.txt
