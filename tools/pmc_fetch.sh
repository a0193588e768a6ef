#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes on micro-benchmark binaries: tools/pmc_fetch.sh <binary> [<binary> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for BIN in "$@"; do
  n=$(basename $BIN)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pf_$n_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pf_${n}_$c -o x -- $R/$BIN 24 300 > /dev/null 2>&1
    python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pf_${n}_$c/**/x_counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "conv5x5" in r["Kernel_Name"] and r["Counter_Name"] == "$c":
        agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    big = [x for x in v if x > 0.5 * max(v)]          # the L=300 launches
    print("$n $c", k, "n=%d mean of large launches = %.0f KiB" % (len(big), sum(big) / len(big)))
PY
  done
done
