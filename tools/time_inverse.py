#!/usr/bin/env python
"""GPU box: dmp_spd_inverse at D = 21 L (the covariance inverse of fast_dca), ms per call and the residual."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dmpfold2_amd import synth      # noqa: E402
from abi import Stages              # noqa: E402

st = Stages(synth.synth_weights(0, coord_scale=5.0), max_L=500, max_N=8)
# trailing update: 0 = operands from L2, 1 = panels staged in LDS, 2 = + tile fetched first (option gj_lds)
modes = [int(x) for x in os.environ.get("GJ_LDS", "0").split(",")]
if os.environ.get("GJ_DIAG"):       # row groups of the diagonal sweep: 2 (256 threads), 4 (512), 8 (1024)
    st.eng.set_option("gj_diag_groups", int(os.environ["GJ_DIAG"]))
    print("gj_diag_groups =", os.environ["GJ_DIAG"], flush=True)
for L in [int(x) for x in sys.argv[1:]] or [82, 200, 300, 500]:
    D = 21 * L
    g = torch.Generator(device="cuda").manual_seed(L)
    B = torch.randn(D, 2 * D, device="cuda", generator=g)
    A = (B @ B.t() / (2 * D) + 0.5 * torch.eye(D, device="cuda")).contiguous()
    first = None
    for mode in modes:
        st.eng.set_option("gj_lds", mode)
        inv = st.spd_inverse(A)
        res = float((A.double() @ inv.double() - torch.eye(D, device="cuda", dtype=torch.float64)).abs().max())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        work = A.clone()
        e0.record()
        for _ in range(5):
            work.copy_(A)
            st.call("dmp_spd_inverse", work, D)
        e1.record()
        torch.cuda.synchronize()
        same = bool(torch.equal(work, inv))
        if first is None:
            first = inv.clone()
        print("D=%d gj_lds=%d: %.3f ms per inverse (incl. a %d MB copy), max|A inv - I| = %.2e, deterministic %s, "
              "equal to the first mode bit for bit %s"
              % (D, mode, e0.elapsed_time(e1) / 5, D * D * 4 // 1000000, res, same, bool(torch.equal(inv, first))), flush=True)
st.eng.sync_check()
