#!/bin/bash
# GPU box: the round-6 profile set.  Kernel stats of the bench command per arithmetic leg (three tables: the headline leg's
# must hold no split-f16 kernel, the float32 leg's no 16-bit matrix-core kernel), PMC passes of the three convolution
# kernels, single-target timeline, lane trace, the bench itself, every BASELINE configuration; summaries are copied to
# profiles/ by tools/summarize_profiles.py r06.  Every step under its own timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06prof; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
for leg in bf16x3 f32 f16x2; do
  rm -rf $R/gpurun_out/prof_bench_$leg; mkdir -p $R/gpurun_out/prof_bench_$leg
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_$leg -o bench -- \
    python $R/bench.py --steps 2 --warmup 1 --legs $leg > $R/gpurun_out/prof_bench_$leg/bench_under_rocprof.log 2>&1
  echo "stats $leg rc=$?"
done
find $R/gpurun_out/prof_bench_* -name "*kernel_trace.csv" -size +20M -delete
for m in 2 1 0; do timeout 900 bash $R/tools/pmc_conv.sh $m 300 m$m > $O/pmc_m$m.txt 2>&1; tail -3 $O/pmc_m$m.txt; done
for p in 2 0; do
  rm -rf /tmp/single_prof
  DMPFOLD_PRECISION=$p timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/single_prof -o s -- python $R/tools/single_trace.py run 300 2000 10 100 4 > $O/single_run_p$p.txt 2>&1
  f=$(find /tmp/single_prof -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && timeout 120 python $R/tools/single_trace.py analyse "$f" > $O/single_timeline_p$p.txt 2>&1
  DMPFOLD_PRECISION=$p timeout 300 python $R/tools/single_trace.py run 300 2000 10 100 6 > $O/single_run_noprof_p$p.txt 2>&1
done
cd $R
DMPFOLD_PRECISION=2 timeout 600 python tools/lane_trace.py > $O/lane_trace_p2.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 1500 python tools/bench_configs.py --skip-c5-f32 > $O/configs.jsonl 2> $O/configs.err
VGRU_F32=1 timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $O/vgru_f32.txt 2>&1
VGRU_F32=2 timeout 600 python tools/time_vgru_persist.py 8 300 2000 > $O/vgru_x3.txt 2>&1
tail -c 600 $O/bench.json; tail -4 $O/single_run_noprof_p2.txt; head -30 $O/single_timeline_p2.txt
