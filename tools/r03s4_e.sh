#!/bin/bash
# GPU box, round 3 session 4, call E: sequence GRU / minimiser cluster with the placement check and XCD-local publication
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4e; mkdir -p $O; cd $R
python tools/time_seq_gru.py 300 > $O/seq.txt 2>&1; cat $O/seq.txt
python tools/time_refine.py > $O/refine.txt 2>&1; tail -6 $O/refine.txt
timeout 300 python tools/single_trace.py run 300 2000 10 100 5 > $O/single.txt 2>&1; tail -3 $O/single.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
