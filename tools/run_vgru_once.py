#!/usr/bin/env python
"""GPU box: one vertical-GRU run for profilers.  python tools/run_vgru_once.py L N K legacy|group"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmpfold2_amd import synth, _lib
from dmpfold2_amd.predict import Engine, encode_aln
L, N, K, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
engs = []
for i in range(K):
    e = Engine(dev, L, N)
    e.set_weights(sd)
    e.set_option("vgru_legacy", 1 if mode == "legacy" else 0)
    engs.append(e)
lib = engs[0].lib
msas = [torch.from_numpy(encode_aln(synth.synth_msa(L, N, 3 + i))).to(dev) for i in range(K)]
outs = [torch.empty(L, 512, device=dev) for _ in range(K)]
for rep in range(2):
    if mode == "legacy":
        for i in range(K):
            _lib.check(lib.dmp_gru_vertical(engs[i].ctx, msas[i].data_ptr(), N, L, outs[i].data_ptr(), engs[i].stream()))
    else:
        ctxs = (C.c_void_p * K)(*[e.ctx for e in engs])
        mp = (C.c_void_p * K)(*[m.data_ptr() for m in msas])
        op = (C.c_void_p * K)(*[o.data_ptr() for o in outs])
        Ns, Ls = (C.c_int * K)(*([N] * K)), (C.c_int * K)(*([L] * K))
        _lib.check(lib.dmp_gru_vertical_group(ctxs, K, mp, Ns, Ls, op, engs[0].stream()))
    torch.cuda.synchronize()
print("ok", bool(torch.isfinite(outs[0]).all()))
