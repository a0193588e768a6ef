#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02e
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-baseline none --no-exact-f32 > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02e/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "chip_ms", d["roofline"]["chip_ms_per_launch"], "avg_launch", d["roofline"]["avg_launch_ms"], "verify", d["verify"])
PY
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/r02e/pytest.log 2>&1
tail -15 gpurun_out/r02e/pytest.log
