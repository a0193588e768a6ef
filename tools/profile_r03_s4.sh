#!/bin/bash
# GPU box: the final round-3 set after session 4 (GPU suite, kernel stats of the bench command, conv PMC passes,
# single-target timeline with and without the profiler, lane trace, the bench itself)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4final; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
bash $R/tools/profile_bench.sh > $O/profile_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/single_prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/single_prof -o s -- python $R/tools/single_trace.py run 300 2000 10 100 4 > $O/single_run.txt 2>&1
f=$(find /tmp/single_prof -name "*kernel_trace.csv" | head -1)
python $R/tools/single_trace.py analyse $f > $O/single_timeline.txt 2>&1
python $R/tools/single_trace.py run 300 2000 10 100 6 > $O/single_run_noprof.txt 2>&1
cd $R; python tools/lane_trace.py > $O/lane_trace.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 700 $O/bench.json; tail -4 $O/single_run_noprof.txt; head -24 $O/single_timeline.txt
