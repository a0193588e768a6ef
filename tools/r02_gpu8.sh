#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02h
out=gpurun_out/r02h/ab.txt
{
for v in tap eo eo168; do echo "== $v one stream"; ./tools/_bin/ubench_conv_sus_$v 300 12 1 | tail -1; done
for v in tap eo168; do echo "== $v two streams"; ./tools/_bin/ubench_conv_sus_$v 300 12 2 | tail -1; done
bash tools/ab_bench.sh dmpfold2_amd/libdmpfold_hip_tap.so dmpfold2_amd/libdmpfold_hip.so 2
} > $out 2>&1
cat $out
