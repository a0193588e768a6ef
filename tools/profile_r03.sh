#!/bin/bash
# GPU box: the round-3 profile set (kernel stats of the bench command, conv PMC passes, single-target timeline, lane trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/profile_bench.sh > $R/gpurun_out/profile_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/single_prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/single_prof -o s -- python $R/tools/single_trace.py run 300 2000 10 100 4 > $R/gpurun_out/single_run.txt 2>&1
f=$(find /tmp/single_prof -name "*kernel_trace.csv" | head -1)
python $R/tools/single_trace.py analyse $f > $R/gpurun_out/single_timeline.txt 2>&1
python $R/tools/single_trace.py run 300 2000 10 100 6 > $R/gpurun_out/single_run_noprof.txt 2>&1
cd $R; python tools/lane_trace.py > gpurun_out/lane_trace.txt 2>&1
python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
tail -c 600 gpurun_out/bench_r03.json; tail -5 gpurun_out/single_run_noprof.txt; head -30 gpurun_out/single_timeline.txt
