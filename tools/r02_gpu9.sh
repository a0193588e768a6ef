#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02i
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/r02i/pytest.log 2>&1
tail -8 gpurun_out/r02i/pytest.log
run() { env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline none --no-exact-f32 --streams $S 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', 'S=$S', 'value %.3f' % d['value'], 'chip_ms %.4f' % r['chip_ms_per_launch'], 'in_flight %.2f' % r['launches_in_flight'], 'ok', d['verify']['ok'])"; }
{
for S in 4 5 6; do for D in 1 2 3; do run DMP_LANE_DEPTH=$D; done; done
S=4; run DMP_LANE_DEPTH=2
} > gpurun_out/r02i/sweep.txt 2>&1
cat gpurun_out/r02i/sweep.txt
