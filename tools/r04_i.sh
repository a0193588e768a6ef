#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
OUT=gpurun_out/r04i; mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/invp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/invp -o inv -- python $R/tools/time_inverse.py 300 > $R/$OUT/prof_pk.txt 2>&1)
grep "^L=" $OUT/prof_pk.txt
f=$(find /tmp/invp -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then n=$(grep -c gj_ "$f" < /dev/null); timeout 60 python tools/inverse_timeline.py "$f" $((n - 90)) 6; fi
