#!/bin/bash
# GPU box: conv kernel variants (pinned LDS prefetch, register cap, software pipeline), alone and in the scheduler
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03d; mkdir -p $O
for v in nopin wpe2nopin wpe2 wpe2pipe; do
  echo "== $v one stream"; timeout 120 tools/_bin/ubench_conv_sus_$v 300 12 1 | tail -4
  echo "== $v two streams"; timeout 120 tools/_bin/ubench_conv_sus_$v 300 12 2 | tail -3
done > $O/conv_sustained.txt 2>&1
cat $O/conv_sustained.txt
for v in wpe2 wpe2pipe; do
  echo "== parity $v"
  DMPFOLD_HIP_LIB=$R/tools/_bin/libconv_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_block or conv_paths or odd_length or scheduler_equals" 2>&1 | tail -3
done > $O/conv_parity.txt 2>&1
cat $O/conv_parity.txt
run() { lib=$1; shift
  out=$(env DMPFOLD_HIP_LIB=$R/$lib "$@" python bench.py --no-cpu-baseline --no-exact-f32 --steps 4 --warmup 1 2>/dev/null)
  python3 - "$lib $*" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-60s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f frac %.3f" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-300:])
PY
}
for i in 1 2; do
  for v in nopin wpe2nopin wpe2 wpe2pipe; do run tools/_bin/libconv_$v.so; done
done > $O/conv_ab_bench.txt 2>&1
cat $O/conv_ab_bench.txt
{ run tools/_bin/libconv_nopin.so DMP_VGRU_DETACH=1 DMP_VGRU_CHAIN_PRIO=1; run tools/_bin/libconv_nopin.so DMP_VGRU_DETACH=1 DMP_VGRU_CHAIN_PRIO=1 GPU_MAX_HW_QUEUES=12; } > $O/detach_prio.txt 2>&1
cat $O/detach_prio.txt
timeout 900 python tools/design_coord_fc.py --out $O/coord_fc > $O/design.log 2>&1; cat $O/design.log
