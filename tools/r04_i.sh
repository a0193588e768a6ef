#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r04i; mkdir -p $OUT
timeout 600 python tools/time_inverse.py 300 500 1000 > $OUT/inverse.txt 2>&1
grep "^L=" $OUT/inverse.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "inverse or dca or cov or features" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 900 python bench.py --steps 2 --warmup 1 --legs f16x3 > $OUT/bench.json 2> $OUT/bench.err
timeout 60 python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['single_target']['ms'], d['verify']['digest_match'], d['verify']['digest'])
PY
