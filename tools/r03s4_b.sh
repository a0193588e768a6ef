#!/bin/bash
# GPU box, round 3 session 4, call B: riders in the group chain (chain of eight), A/B of the bench, eigen-gap screen at
# L = 1344 with deeper alignments, kernel timeline of one prediction
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grouped_vertical or group_stage or pipeline_matches" > $O/pytest_group.txt 2>&1; tail -4 $O/pytest_group.txt
timeout 300 python tools/time_vgru_group.py 8 300 2000 > $O/vgru_group8.txt 2>&1; tail -12 $O/vgru_group8.txt
STEPS=4 timeout 900 bash tools/ab_env.sh DMP_VGRU_RIDERS 0 4 2 > $O/ab_riders.txt 2>&1; cat $O/ab_riders.txt
timeout 300 python bench.py --no-cpu-baseline --no-exact-f32 > $O/bench_quick.json 2> $O/bench_quick.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/s4b/bench_quick.json").read().strip().splitlines()[-1])
print("bench", j["value"], "digest", j["verify"].get("digest"), "ok", j["verify"]["ok"], {k: v for k, v in j["verify"].items() if k.startswith("reference")})
PY
timeout 900 python tools/screen_eig_gaps.py --L 1344 --N 1000 --seeds 0-11 --n 0 > $O/gaps_1344_N1000.txt 2> $O/gaps.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/single_prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/single_prof -o s -- python $R/tools/single_trace.py run 300 2000 10 100 4 > $O/single_run.txt 2>&1
f=$(find /tmp/single_prof -name "*kernel_trace.csv" | head -1)
python $R/tools/single_trace.py analyse $f > $O/single_timeline.txt 2>&1; head -14 $O/single_timeline.txt
