#!/bin/bash
# GPU box: the round-5 profile set.  Kernel stats of the bench command per arithmetic leg (two tables: the float32 leg's must
# hold no f16 / bf16 matrix-core kernel), PMC passes of both convolution kernels, single-target timeline, lane trace, the
# timings of the stages that changed this round, the bench itself; summaries are copied to profiles/ by
# tools/summarize_profiles.py r05.  Every step under its own timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05prof; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/prof_bench_f32 $R/gpurun_out/pmc /tmp/single_prof
mkdir -p $R/gpurun_out/prof_bench $R/gpurun_out/prof_bench_f32 $R/gpurun_out/pmc
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- \
  python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-files-leg --legs f16x3 > $R/gpurun_out/prof_bench/bench_under_rocprof.log 2>&1
echo "stats f16x3 rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_f32 -o bench -- \
  python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-files-leg --legs f32 > $R/gpurun_out/prof_bench_f32/bench_under_rocprof.log 2>&1
echo "stats f32 rc=$?"
run() { mode=$1; name=$2; shift; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/tools/conv_only.py 1 300 $mode > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
for mode in 0 1; do
  sfx=$([ $mode = 1 ] && echo _f32 || echo "")
  run $mode sq1$sfx SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
  run $mode sq2$sfx SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  run $mode fetch$sfx FETCH_SIZE
  run $mode write$sfx WRITE_SIZE
done
find $R/gpurun_out/prof_bench $R/gpurun_out/prof_bench_f32 $R/gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/single_prof -o s -- python $R/tools/single_trace.py run 300 2000 10 100 4 > $O/single_run.txt 2>&1
f=$(find /tmp/single_prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && timeout 120 python $R/tools/single_trace.py analyse "$f" > $O/single_timeline.txt 2>&1
timeout 300 python $R/tools/single_trace.py run 300 2000 10 100 6 > $O/single_run_noprof.txt 2>&1
cd $R; timeout 600 python tools/lane_trace.py > $O/lane_trace.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $O/vgru_persist.txt 2>&1
VGRU_F32=1 timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $O/vgru_f32.txt 2>&1
timeout 600 python tools/time_inverse.py 300 500 1000 > $O/inverse.txt 2>&1
timeout 600 python tools/time_bwd.py 300 350 > $O/bwd_time.txt 2>&1
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
timeout 1500 python tools/parity_budget.py > $O/parity_budget.txt 2> $O/parity_budget.err
timeout 300 python tools/batch_throughput.py 64 300 2000 > $O/batch_throughput.txt 2>&1
tail -c 600 $O/bench.json; tail -4 $O/single_run_noprof.txt; head -30 $O/single_timeline.txt
