#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $OUT/vgru_persist.txt 2>&1
echo "vgru rc=$?" >> $OUT/vgru_persist.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
DMP_WRITE_DIGEST=1 timeout 900 python bench.py --steps 10 --warmup 2 --cpu-baseline none > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
cp profiles/bench_digest.json $OUT/bench_digest.json
tail -12 $OUT/vgru_persist.txt; tail -3 $OUT/tests.log; cut -c1-600 $OUT/bench.json
