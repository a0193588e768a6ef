#!/bin/bash
# round 2, GPU call 1: whole GPU suite, bench with all legs, scheduler idle-sleep A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
(time timeout 900 python bench.py --steps 10 --warmup 2) > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
tail -c 3000 gpurun_out/r02a/bench.json
for us in 0 20 100; do
  DMP_PUMP_SLEEP_US=$us timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline none --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sleep_us=$us', d['value'], d['roofline']['frac'])" >> gpurun_out/r02a/pump_sleep.txt
done
cat gpurun_out/r02a/pump_sleep.txt
