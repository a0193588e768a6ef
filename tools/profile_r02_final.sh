#!/bin/bash
# GPU box: PMC passes of the convolution + kernel stats of the bench + GPU suite at the final state of round 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02v
rm -rf $O $R/gpurun_out/prof_bench $R/gpurun_out/pmc; mkdir -p $O $R/gpurun_out/prof_bench $R/gpurun_out/pmc
python $R/bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- \
  python $R/bench.py --steps 2 --warmup 1 --cpu-baseline none --no-exact-f32 > $R/gpurun_out/prof_bench/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
mkdir -p $O/depth1
DMP_LANE_DEPTH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/depth1 -o bench -- \
  python $R/bench.py --steps 2 --warmup 1 --cpu-baseline none --no-exact-f32 > $O/depth1/bench_under_rocprof.log 2>&1
find $O/depth1 -name "*kernel_trace.csv" -delete
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/tools/conv_only.py 5 300 > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
find $R/gpurun_out/prof_bench $R/gpurun_out/pmc -name "*kernel_trace.csv" -size +20M -delete
cd $R && (time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
grep "passed\|failed" $O/pytest.log; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['verify']['ok'], d['verify']['digest_match'], d['cpu_baseline']['seconds_per_structure'])"
