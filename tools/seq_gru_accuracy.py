#!/usr/bin/env python
"""GPU box: error of the sequence GRUs (hgru: 2 layers, coord_gru: 3 layers, T = 300) against the SAME recurrence in
float64 (torch.nn.GRU with float64 weights on the CPU), and of torch's own float32 CPU GRU against it - the yardstick
for the gate functions (device library / hardware forms).   [DMPFOLD_HIP_LIB=...] python tools/seq_gru_accuracy.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmpfold_oracle as O                       # noqa: E402
from dmpfold2_amd import synth                   # noqa: E402
from abi import Stages                           # noqa: E402

L = 300


def gru(weights, prefix, x, nin, layers, bf, dtype):
    g = torch.nn.GRU(nin, 256, num_layers=layers, bidirectional=True, batch_first=bf).to(dtype)
    g.load_state_dict({k[len(prefix) + 1:]: v.to(dtype) for k, v in weights.items() if k.startswith(prefix + ".")})
    g.eval()
    with torch.no_grad():
        return g(x.to(dtype))[0]


sd = synth.synth_weights(0, coord_scale=5.0)
W32 = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
W64 = {k: v.double() for k, v in W32.items()}
st = Stages(sd, max_L=L, max_N=8)
g = torch.Generator().manual_seed(5)
for name, which, prefix, nin, layers, scale in (("hgru", 0, "hgru", 512, 2, 0.5), ("coord_gru", 1, "coord_gru", 520, 3, 1.0)):
    errs = []
    for rep in range(3):
        x = torch.randn(L, nin, generator=g) * scale
        got = st.gru_bidir(which, x.cuda()).cpu().double()
        bf = prefix == "coord_gru"
        xin = x.unsqueeze(0) if bf else x.unsqueeze(1)
        r64 = gru(W32, prefix, xin, nin, layers, bf, torch.float64).reshape(L, 512)
        r32 = gru(W32, prefix, xin, nin, layers, bf, torch.float32).reshape(L, 512).double()
        errs.append(((got - r64).abs().max().item(), (got - r64).pow(2).mean().sqrt().item(),
                     (r32 - r64).abs().max().item(), (r32 - r64).pow(2).mean().sqrt().item()))
    e = np.array(errs).mean(0)
    print("%-9s HIP vs float64: max %.2e rms %.2e | torch float32 CPU vs float64: max %.2e rms %.2e" % (name, *e), flush=True)
