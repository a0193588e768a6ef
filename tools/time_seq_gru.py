#!/usr/bin/env python
"""GPU box: the sequence GRUs alone (hgru: 2 layers, coord_gru: 3 layers; T = L steps each), ms per call, ms per layer
and us per step, with a SHA-256 of the outputs (variants built with DMP_FLAGS_GRU must reproduce the default's bits).

    [DMPFOLD_HIP_LIB=tools/_bin/libseq_xxx.so] python tools/time_seq_gru.py [L=300]
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dmpfold2_amd import synth      # noqa: E402
from abi import Stages              # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 300
st = Stages(synth.synth_weights(0, coord_scale=5.0), max_L=L, max_N=8)
g = torch.Generator(device="cuda").manual_seed(1)
v = torch.randn(L, 512, device="cuda", generator=g) * 0.3
emb = torch.randn(L, 520, device="cuda", generator=g) * 0.3
for which, x, layers in ((0, v, 2), (1, emb, 3)):
    out = st.gru_bidir(which, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        st.gru_bidir(which, x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%s L=%d: %.3f ms per call = %.3f ms per layer (incl. its input-projection GEMM) = %.2f us per step; sha %s"
          % ("hgru" if which == 0 else "coord_gru", L, ms, ms / layers, ms / layers / L * 1e3,
             hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]), flush=True)
st.eng.sync_check()
