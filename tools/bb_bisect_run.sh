#!/bin/bash
# GPU box: run tools/bb_hazard over the variants of tools/bb_bisect.py; prints one line per variant
K=_ZN3dmp15backbone_kernelEPKfS1_iffPfS2_
N=${N:-3000}
for f in tools/_bin/bbv/${1:-only_}*.hsaco tools/_bin/bbv/all_scalar.hsaco tools/_bin/bbv/v0.hsaco; do
  r=$(timeout 60 tools/_bin/bb_hazard $f $K $N | tr '\n' ' ' | sed 's/launches differ from the reference; residues by wave quarter/;q/g; s/max |diff|/max/g')
  echo "$(basename $f .hsaco): $r"
done
