#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout (a step that
# waited on an empty argument once cost a whole GPU call); outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/all.log 2>&1; tail -3 $OUT/all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json
