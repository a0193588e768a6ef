#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; OUT=gpurun_out/job; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "inverse or covariance" > $OUT/inv_tests.log 2>&1; tail -3 $OUT/inv_tests.log
timeout 600 python tools/time_inverse.py 300 500 > $OUT/inverse.txt 2>&1; grep "pairs \|chain128\|bitwise" $OUT/inverse.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/invprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/invprof -o inv -- python $R/tools/time_inverse.py 300 > /dev/null 2>&1
f=$(find /tmp/invprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$OUT/inv_kernel_stats.csv && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gj_" in r["Name"]:
        print("%-40s calls %5s avg %8.1f us" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
