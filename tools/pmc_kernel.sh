#!/bin/bash
# PMC passes for one kernel of tools/gpu_diag.py --time: tools/pmc_kernel.sh <kernel-substring> <counter> [<counter> ...]
# (one rocprofv3 run per counter; prints the mean per launch over the launches of that kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/pk_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pk_$c -o x -- python $R/tools/gpu_diag.py --time 300 2000 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pk_$c/**/x_counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "$K" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
v.sort()
print("$K $c: n=%d median=%.6g mean=%.6g" % (len(v), v[len(v)//2] if v else 0, sum(v)/max(1,len(v))))
PY
done
