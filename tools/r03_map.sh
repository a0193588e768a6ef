#!/bin/bash
# GPU box: block -> (tile, split) maps of the convolution: sustained time and L2-fabric traffic
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r03map; mkdir -p $O
for v in nopin map1 map2; do echo "== $v"; timeout 120 tools/_bin/ubench_conv_sus_$v 300 10 1 | tail -1; timeout 120 tools/_bin/ubench_conv_sus_$v 300 10 2 | tail -1; done > $O/sustained.txt 2>&1
cat $O/sustained.txt
cd /tmp && export TMPDIR=/tmp
for v in nopin map1 map2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$v$c
    DMPFOLD_HIP_LIB=$R/tools/_bin/libconv_$v.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$v$c -o x -- python $R/tools/conv_only.py 5 300 > /tmp/pm.log 2>&1
    python3 - $v $c <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/pm_{sys.argv[1]}{sys.argv[2]}/**/x_counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "conv5x5_f16x3" in r["Kernel_Name"]] if f else []
print(sys.argv[1], sys.argv[2], "n=%d mean=%.0f KiB" % (len(v), sum(v) / max(1, len(v))))
PY
  done
done > $O/traffic.txt 2>&1
cat $O/traffic.txt
DMPFOLD_HIP_LIB=$R/tools/_bin/libconv_map1.so timeout 300 python -m pytest $R/tests/test_gpu_parity.py -q -x -k "test_block or odd_length" 2>&1 | tail -1
DMPFOLD_HIP_LIB=$R/tools/_bin/libconv_map2.so timeout 300 python -m pytest $R/tests/test_gpu_parity.py -q -x -k "test_block or odd_length" 2>&1 | tail -1
