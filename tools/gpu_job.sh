#!/bin/bash
# scratch GPU job of a session: gpurun -- 'bash tools/gpu_job.sh'.  Every step under its own timeout (a step that
# waited on an empty argument once cost a whole GPU call); outputs under gpurun_out/job/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/job; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/all.log 2>&1; tail -25 $OUT/all.log
timeout 1500 python bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; tail -5 $OUT/bench.err
