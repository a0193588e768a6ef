#!/bin/bash
# build debug variants of the vertical-GRU kernel (CPU side) ...   tools/vg_variants.sh build
# ... and time them on the GPU box                                 tools/vg_variants.sh run
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  mkdir -p $R/tools/_bin
  for v in ${VG_VARIANTS:-NOMFMA SAMEADDR NOEPI "NOMFMA -DVG_DBG_SAMEADDR" NOLOOP NOTHING}; do
    tag=$(echo $v | tr -d ' -' ); 
    DMP_FLAGS_VGRU="-DVG_DBG_$v" DMP_LIB_OUT=$R/tools/_bin/libvg_$tag.so python -m dmpfold2_amd.build --force > /dev/null || exit 1
    echo built $tag
  done
  python -m dmpfold2_amd.build --force > /dev/null
else
  for f in $R/tools/_bin/libvg_*.so; do
    echo "== $(basename $f)"; DMPFOLD_HIP_LIB=$f python $R/tools/time_vgru_group.py 4 300 2000 2>&1 | grep "ONE chain\|group kernel, chains"
  done
fi
