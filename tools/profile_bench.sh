#!/bin/bash
# GPU box: kernel-trace stats of the benchmark command + PMC passes of the conv kernel.
# Outputs under gpurun_out/prof_bench and gpurun_out/pmc; copy summaries to profiles/ afterwards
# (tools/summarize_profiles.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc
mkdir -p $R/gpurun_out/prof_bench $R/gpurun_out/pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- \
  python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/tools/conv_only.py 5 300 > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
# keep only the small files
find $R/gpurun_out/prof_bench $R/gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
ls -la $R/gpurun_out/prof_bench/* | head; du -sh $R/gpurun_out
