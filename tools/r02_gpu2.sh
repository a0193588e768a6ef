#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02b
(time timeout 1200 python -m pytest tests/test_gpu_headline.py tests/test_gpu_fullsize.py -m gpu -q) > gpurun_out/r02b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02b/pytest.log
tail -30 gpurun_out/r02b/pytest.log
python tools/perpass_dev.py pf10963_n10_m0 fit3fgx_L96_N50_n10_m100 fit3fgx_L96_N50_n0_m100 > gpurun_out/r02b/perpass.txt 2>&1
cat gpurun_out/r02b/perpass.txt
