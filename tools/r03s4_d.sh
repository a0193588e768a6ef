#!/bin/bash
# GPU box, round 3 session 4, call D: where a step of the sequence GRU goes (cycle stamps), variants of its reduction
# and publication
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4d; mkdir -p $O; cd $R
( echo "== default (DPP reduction, agent-scope store)"; python tools/time_seq_gru.py 300
  for v in shfl xcd; do echo "== $v"; DMPFOLD_HIP_LIB=$R/tools/_bin/libseq_$v.so python tools/time_seq_gru.py 300; done
  for v in prof profshfl profxcd; do echo "== $v"; DMPFOLD_HIP_LIB=$R/tools/_bin/libseq_$v.so python tools/time_seq_gru.py 300 | sort | uniq -c | sort -rn | head -12; done ) > $O/seq.txt 2>&1
cat $O/seq.txt
