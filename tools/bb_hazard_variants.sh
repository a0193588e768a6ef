#!/bin/bash
# Build code objects with the backbone kernel for tools/bb_hazard: the kernel as compiled, without the
# SLP vectoriser (no packed-f32 instructions), and with its assembly edited (idle cycles around packed ops).
# Runs in the build container (hipcc cross-compiles); output: tools/_bin/bbv/*.hsaco
R=$(cd "$(dirname "$0")/.." && pwd)
LL=/opt/rocm/lib/llvm/bin
O=$R/tools/_bin/bbv
mkdir -p $O
dev_asm() {   # $1 = output .s, rest = extra flags
  out=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -S --cuda-device-only -I$R/dmpfold2_amd/csrc -I$R/include \
        $R/dmpfold2_amd/csrc/coords.hip -o $out 2>/dev/null
}
to_hsaco() {  # $1 = .s, $2 = .hsaco
  $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $1 -o ${1%.s}.o && $LL/ld.lld -shared ${1%.s}.o -o $2
}
dev_asm $O/v0.s
dev_asm $O/noslp.s -fno-slp-vectorize
sed 's/^\(\s*v_pk_.*\)$/\1\n\ts_nop 7/' $O/v0.s > $O/nop_after_pk.s
sed 's/^\(\s*v_pk_.*\)$/\ts_nop 7\n\1/' $O/v0.s > $O/nop_before_pk.s
sed 's/^\(\s*v_pk_.*\)$/\ts_nop 7\n\1\n\ts_nop 7/' $O/v0.s > $O/nop_around_pk.s
for extra in "$@"; do [ -f "$extra" ] && cp "$extra" $O/; done
for f in $O/*.s; do to_hsaco $f ${f%.s}.hsaco && echo "built ${f%.s}.hsaco ($(grep -c 'v_pk_' $f) packed ops, $(grep -c 's_nop 7' $f) added idle slots)"; done
rm -f $O/*.o
