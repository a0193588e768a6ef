#!/bin/bash
# scratch GPU job of the session (gpurun -- 'bash tools/gpu_job.sh'); every step under a timeout
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "grouped_vertical" > $OUT/t$i.log 2>&1
tail -2 $OUT/t$i.log
done
grep -n "^E " $OUT/t*.log | head -10
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/all.log 2>&1; tail -3 $OUT/all.log
for rep in 1 2; do
for lib in tools/_bin/libdmp_prev.so dmpfold2_amd/libdmpfold_hip.so; do
  echo "== $lib"
  DMPFOLD_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 4 --warmup 1 --legs f16x3 --no-cpu-baseline 2>/dev/null | timeout 60 python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['single_target']['ms'], d['verify']['digest_match'], d['roofline']['chip_ms_per_launch'])"
done
done
