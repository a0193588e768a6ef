#!/usr/bin/env python
"""GPU box: new vertical-GRU step kernel against the legacy one on a few shapes, repeated (race detector)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmpfold2_amd import synth, _lib
from dmpfold2_amd.predict import Engine, encode_aln
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
e = Engine(dev, 320, 2000)
e.set_weights(sd)
lib = e.lib
for (L, N) in [(82, 252), (96, 5), (300, 1), (300, 20), (300, 130), (300, 300), (300, 2000), (320, 500), (33, 64)]:
    msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, 7))).to(dev)
    out = torch.empty(L, 512, device=dev)
    e.set_option("vgru_legacy", 1)
    _lib.check(lib.dmp_gru_vertical(e.ctx, msa.data_ptr(), N, L, out.data_ptr(), e.stream()))
    torch.cuda.synchronize()
    ref = out.clone()
    e.set_option("vgru_legacy", 0)
    res = []
    for rep in range(3):
        out.fill_(7.0)
        _lib.check(lib.dmp_gru_vertical(e.ctx, msa.data_ptr(), N, L, out.data_ptr(), e.stream()))
        torch.cuda.synchronize()
        res.append(out.clone())
    fin = [bool(torch.isfinite(r).all()) for r in res]
    dif = [float((r - ref).abs().max()) if f else float("nan") for r, f in zip(res, fin)]
    same = all(torch.equal(res[0], r) for r in res[1:])
    bad_cols = sorted(set(torch.nonzero(~torch.isfinite(res[0]).all(dim=1)).flatten().tolist()))[:12]
    print(f"L={L} N={N}: finite {fin} max|new-legacy| {dif} reproducible {same} bad rows(first) {bad_cols}", flush=True)
