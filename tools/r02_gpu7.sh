#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02g
out=gpurun_out/r02g/sustained.txt
: > $out
smi() { while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Power (W)\|Socket Power" | tr '\n' ' ' ; echo; sleep 0.5; done; }
for v in tap eo tap eo; do
  echo "== $v, one stream" >> $out
  smi > gpurun_out/r02g/smi_$v.txt & SP=$!
  ./tools/_bin/ubench_conv_sus_$v 300 25 1 >> $out 2>&1
  kill $SP
done
for v in tap eo; do
  echo "== $v, two streams" >> $out
  ./tools/_bin/ubench_conv_sus_$v 300 25 2 >> $out 2>&1
done
cat $out
tail -3 gpurun_out/r02g/smi_eo.txt
