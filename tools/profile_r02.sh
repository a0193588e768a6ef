#!/bin/bash
# GPU box, round 2: bench with every leg + digest, rocprofv3 kernel stats of the bench at lane depth 2 (the
# default) and 1 (one convolution at a time: roofline.frac is recomputable from its AverageNs alone), PMC
# passes of the convolution, timeline of one single-target prediction.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02p
rm -rf $O $R/gpurun_out/prof_bench $R/gpurun_out/pmc; mkdir -p $O $R/gpurun_out/prof_bench $R/gpurun_out/pmc
DMP_WRITE_DIGEST=1 python $R/bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cp $R/profiles/bench_digest.json $O/ 2>/dev/null
python $R/bench.py --steps 10 --warmup 2 --cpu-baseline none --no-exact-f32 > $O/bench_repeat.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- \
  python $R/bench.py --steps 2 --warmup 1 --cpu-baseline none --no-exact-f32 > $R/gpurun_out/prof_bench/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
mkdir -p $O/depth1
DMP_LANE_DEPTH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/depth1 -o bench -- \
  python $R/bench.py --steps 2 --warmup 1 --cpu-baseline none --no-exact-f32 > $O/depth1/bench_under_rocprof.log 2>&1
echo "depth1 rc=$?"
find $O/depth1 -name "*kernel_trace.csv" -delete
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/tools/conv_only.py 5 300 > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
find $R/gpurun_out/prof_bench $R/gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
python $R/tools/single_trace.py run 300 2000 10 100 4 > $O/latency.txt 2>&1
rm -rf /tmp/st && rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o st -- python $R/tools/single_trace.py run 300 2000 10 100 3 > $O/under_rocprof.txt 2>&1
python $R/tools/single_trace.py analyse $(find /tmp/st -name "*kernel_trace.csv" | head -1) > $O/timeline.txt 2>&1
cd $R && (time timeout 1200 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; cat $O/latency.txt; head -c 1500 $O/bench.json; du -sh $R/gpurun_out
