#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02l
smi() { while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/.*: //' | tr '\n' ' ' ; echo; sleep 0.5; done; }
smi > gpurun_out/r02l/smi.txt & SP=$!
./tools/_bin/ubench_mfma_power 2 > gpurun_out/r02l/mfma_power.txt 2>&1
./tools/_bin/ubench_mfma_power 1 >> gpurun_out/r02l/mfma_power.txt 2>&1
kill $SP
cat gpurun_out/r02l/mfma_power.txt
awk 'NR%4==0' gpurun_out/r02l/smi.txt | head -20
