#!/bin/bash
# GPU box: PMC passes (separate rocprofv3 --pmc runs) of the convolution kernel of one arithmetic setting, one trunk pass at
# length L (16 launches): tools/pmc_conv.sh <conv_mode 0|1|2> [L] [tag]  -> gpurun_out/pmc/<tag>*.  Prints the per-launch means.
R=${GRAFT_REPO_ROOT:-/root/repo}; mode=${1:-2}; L=${2:-300}; tag=${3:-m$mode}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
run() { name=$1; shift; rm -rf $R/gpurun_out/pmc/$name; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/tools/conv_only.py 1 $L $mode > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run ${tag}_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run ${tag}_fetch FETCH_SIZE
run ${tag}_write WRITE_SIZE
find $R/gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc/${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv5x5" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]; print("%-28s n=%d mean=%.6g min=%.6g max=%.6g" % (k, len(v), sum(v)/len(v), min(v), max(v)))
if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "GRBM_GUI_ACTIVE" in agg:
    b = sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"])/len(agg["SQ_VALU_MFMA_BUSY_CYCLES"]); g = sum(agg["GRBM_GUI_ACTIVE"])/len(agg["GRBM_GUI_ACTIVE"])
    print("# MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs) = %.3f" % (b / (1024.0 * g / 8.0)))
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    f = sum(agg["FETCH_SIZE"])/len(agg["FETCH_SIZE"]); w = sum(agg["WRITE_SIZE"])/len(agg["WRITE_SIZE"])
    print("# L2-fabric bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB = %.1f MB" % ((2*f + w) * 1024 / 1e6))
PY
