#!/bin/bash
# GPU box: kernel trace of the scheduler (8 targets) and the content of its front-end hole; env is passed through
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-hole}
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/holeprof
rocprofv3 --kernel-trace --output-format csv -d /tmp/holeprof -o t -- python $R/tools/lane_trace.py 4 ${HOLE_TARGETS:-8} > $O/lane_trace_under_rocprof.txt 2>&1
python $R/tools/hole_profile.py /tmp/holeprof > $O/hole_profile.txt 2>&1
cat $O/hole_profile.txt
