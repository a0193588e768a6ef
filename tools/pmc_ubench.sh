#!/bin/bash
# PMC passes on a micro-benchmark binary: tools/pmc_ubench.sh <binary> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcu; mkdir -p $R/gpurun_out/pmcu
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcu/$name -o $name -- $R/$BIN $ARGS > $R/gpurun_out/pmcu/$name.log 2>&1; echo "$name rc=$?"; }
ARGS="$@"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
find $R/gpurun_out/pmcu -name "*kernel_trace.csv" -size +5M -delete
