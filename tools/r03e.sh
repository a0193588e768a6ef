#!/bin/bash
# GPU box: vertical-GRU step kernel variants (pinned LDS prefetch, ring shapes); MFMA-busy of the conv variants; inverse
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r03e; mkdir -p $O
for f in tools/_bin/libvg_*.so; do
  echo "== $(basename $f)"
  DMPFOLD_HIP_LIB=$R/$f timeout 300 python tools/time_vgru_group.py 4 300 2000 2>&1 | grep "ONE chain\|bitwise\|side by side"
  DMPFOLD_HIP_LIB=$R/$f timeout 300 python tools/time_vgru_group.py 1 300 2000 2>&1 | grep "ONE chain"
done > $O/vgru_variants.txt 2>&1
cat $O/vgru_variants.txt
echo "== inverse: old gemm (libconv_nopin) / pinned gemm (default)" > $O/inverse.txt
DMPFOLD_HIP_LIB=$R/tools/_bin/libconv_nopin.so timeout 300 python tools/time_inverse.py 300 500 >> $O/inverse.txt 2>&1
timeout 300 python tools/time_inverse.py 300 500 >> $O/inverse.txt 2>&1
cat $O/inverse.txt
cd /tmp && export TMPDIR=/tmp
pmc() { tag=$1; lib=$2; shift 2
  rm -rf /tmp/pmc_$tag
  DMPFOLD_HIP_LIB=$lib rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o x -- python $R/tools/conv_only.py 5 300 > /tmp/pmc_$tag.log 2>&1
  python3 - $tag <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"/tmp/pmc_{tag}/**/x_counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "conv5x5_f16x3" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-12s %-28s n=%d mean=%.6g" % (tag, k, len(v), sum(v) / len(v)))
PY
}
{
for v in nopin wpe2 wpe2pipe; do
  pmc ${v}_sq1 $R/tools/_bin/libconv_$v.so SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
  pmc ${v}_sq2 $R/tools/_bin/libconv_$v.so SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
done
} > $O/conv_variants_pmc.txt 2>&1
cat $O/conv_variants_pmc.txt
