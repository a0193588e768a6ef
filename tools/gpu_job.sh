#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile_shapes or conv or block" > $OUT/tile_tests.log 2>&1; tail -3 $OUT/tile_tests.log
for m in 2 0; do for L in 300 200 296 104; do timeout 300 python tools/conv_only.py 60 $L $m 2>&1 | tail -1; done; done | tee $OUT/conv_only.log
timeout 900 python bench.py --steps 4 --warmup 1 --no-files-leg --cpu-baseline none > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/job/bench.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','value_f32','value_split_f16')}, d['verify']['ok'], {k:v.get('match') for k,v in d['verify'].items() if k.startswith('digest')}, d['roofline']['chip_ms_per_launch'], d['roofline_split_f16']['chip_ms_per_launch'])
PY
