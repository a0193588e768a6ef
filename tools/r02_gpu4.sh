#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02d
{
RR_EO=1 ./tools/_bin/ubench_conv_rr 24 300
./tools/_bin/ubench_conv_rr 24 300 | grep "ms avg"
RR_EO=1 ./tools/_bin/ubench_conv_rr 24 300 | grep "ms avg"
UB_DATA=zero RR_EO=1 ./tools/_bin/ubench_conv_rr 24 300 | grep "ms avg"
} > gpurun_out/r02d/conv_eo.txt 2>&1
cat gpurun_out/r02d/conv_eo.txt
