#!/bin/bash
# GPU box, round 3 session 4, call C: hand-off latency by cache policy, LDS-staged trailing update of the inverse
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4c; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/ubench_handoff tools/ubench_handoff.hip > $O/handoff_build.txt 2>&1
timeout 300 /tmp/ubench_handoff > $O/handoff.txt 2>&1; cat $O/handoff.txt
GJ_LDS=0,1,2,0,1,2 timeout 600 python tools/time_inverse.py 82 300 500 > $O/inverse.txt 2>&1; cat $O/inverse.txt
