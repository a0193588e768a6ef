#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r02j
python $R/tools/single_trace.py run 300 2000 10 100 4 > $R/gpurun_out/r02j/latency.txt 2>&1
rm -rf /tmp/st && rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o st -- python $R/tools/single_trace.py run 300 2000 10 100 3 > $R/gpurun_out/r02j/under_rocprof.txt 2>&1
f=$(find /tmp/st -name "*kernel_trace.csv" | head -1)
python $R/tools/single_trace.py analyse $f > $R/gpurun_out/r02j/timeline.txt 2>&1
cat $R/gpurun_out/r02j/latency.txt $R/gpurun_out/r02j/timeline.txt
