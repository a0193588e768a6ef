#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4h; mkdir -p $O; cd $R
DMPFOLD_HIP_LIB=$R/tools/_bin/${PROFLIB:-libtc_prof.so} timeout 300 python - > $O/tc_prof.txt 2>&1 <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from dmpfold2_amd import synth
from abi import Stages
st = Stages(synth.synth_weights(0, coord_scale=5.0), max_L=300, max_N=4)
rng = np.random.default_rng(1)
L = 300
P = np.cumsum(rng.standard_normal((L, 3)) * 2.2, axis=0)
D = np.linalg.norm(P[:, None] - P[None], axis=2) + np.abs(rng.standard_normal((L, L))) * 0.3
D = 0.5 * (D + D.T)
M = st.to((0.5 * (D[0:1, :] ** 2 + D[:, 0:1] ** 2 - D ** 2)).astype(np.float32))
for _ in range(2):
    st.eigh_top8(M); torch.cuda.synchronize()
PY
sort $O/tc_prof.txt | uniq | tail -14
