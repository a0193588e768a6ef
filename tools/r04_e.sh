#!/bin/bash
# round 4: full GPU suite + bench with the persistent chain on and off (same box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout 1700 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
DMP_WRITE_DIGEST=1 timeout 900 python bench.py --steps 10 --warmup 2 --cpu-baseline none > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
cp profiles/bench_digest.json $OUT/bench_digest.json
timeout 900 python bench.py --steps 10 --warmup 2 --cpu-baseline none --no-exact-f32 --vgru-per-row > $OUT/bench_rows.json 2> $OUT/bench_rows.err
tail -3 $OUT/tests.log; cut -c1-400 $OUT/bench.json; cut -c1-300 $OUT/bench_rows.json
