#!/bin/bash
# tools/build_variant.sh <name> <file-stem> "<flags>": a library variant with extra flags for ONE source file
# (DMP_FLAGS_<FILE>), written to tools/_bin/libdmp_<name>.so; the object of that file is rebuilt for the default library
# afterwards.  Use with DMPFOLD_HIP_LIB=tools/_bin/libdmp_<name>.so.
set -e
cd "$(dirname "$0")/.."
name=$1; stem=$2; flags=$3
mkdir -p tools/_bin
up=$(echo $stem | tr a-z A-Z)
rm -f dmpfold2_amd/csrc/_build/$stem.o
env DMP_FLAGS_$up="$flags" DMP_LIB_OUT=$PWD/tools/_bin/libdmp_$name.so python -c "import dmpfold2_amd.build as b; b.build(verbose=False)"
rm -f dmpfold2_amd/csrc/_build/$stem.o
