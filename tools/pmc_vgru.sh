#!/bin/bash
# PMC passes over the vertical-GRU step kernels: tools/pmc_vgru.sh <L> <N> <K> <legacy|group> <counter> [...]
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; N=$2; K=$3; M=$4; shift 4
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/pv_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pv_$c -o x -- python $R/tools/run_vgru_once.py $L $N $K $M > /tmp/pv_$c.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pv_$c/**/x_counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "$c"] if f else []
v.sort()
print("L=$L N=$N K=$K $M $c: n=%d median=%.6g mean=%.6g max=%.6g" % (len(v), v[len(v)//2] if v else 0, sum(v)/max(1,len(v)), v[-1] if v else 0))
PY
done
