#!/usr/bin/env python
"""GPU box: stress test for result corruption under co-running kernels.

Runs many short predictions (L=300, N=200, 1 recycling iteration, 100 minimiser steps) through the
multi-engine scheduler and compares every result bit for bit with the single-engine result.  Prints
the number of mismatching results and where they differ (16-residue block, atom indices).
Environment: DBG_TRIALS (20), DBG_S engines (3), DBG_CONV_MODE (0/1/2), DBG_TRI_SINGLE, DBG_N, DBG_IT.
History: 1-4 % of the results used to have the C/O/CB atoms of 16 consecutive residues (lanes 48..63 of a
backbone-kernel wave) wrong when f16 / bf16 convolutions of another target shared the CUs.  Cause: a
packed-f32 instruction form (tools/pk_hazard.hip, tools/bb_hazard.hip reproduce it in
seconds); coords.hip is now compiled without it and tools/isa_lint.py guards the library.
"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dmpfold2_amd import synth
from dmpfold2_amd.predict import Engine, Pipeline, encode_aln
L, N, IT, MS = 300, int(os.environ.get("DBG_N", "200")), int(os.environ.get("DBG_IT", "1")), 100
sdn = synth.synth_weights(0, coord_scale=5.0)
sd = {k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}
msas = [encode_aln(synth.synth_msa(L, N, seed=40 + i)) for i in range(4)]
dev = torch.device("cuda:0")
mode = os.environ.get("DBG_CONV_MODE"); tri = os.environ.get("DBG_TRI_SINGLE")
def opts(e):
    if mode: e.set_option("conv_mode", int(mode))
    if tri: e.set_option("tridiag_single", 1)
e1 = Engine(dev, L, N); e1.set_weights(sd); opts(e1)
refs = []
for m in msas:
    c, f = e1.predict(m, None, IT, MS); e1.sync_check(); refs.append(torch.cat((c.reshape(L, 15), f.reshape(L, 1)), 1).clone())
bad = 0; total = 0; pat = {}
S = int(os.environ.get("DBG_S", "3"))
for trial in range(int(os.environ.get("DBG_TRIALS", "20"))):
    pipe = Pipeline(dev, L, N, sd, streams=S)
    for e in pipe.engines: opts(e)
    order = [i % 4 for i in range(12)]
    res = pipe.run([torch.from_numpy(msas[i]).to(dev) for i in order], IT, MS); pipe.sync_check(); torch.cuda.synchronize()
    for k, i in enumerate(order):
        total += 1
        got = torch.cat((res[k][0].reshape(L, 15), res[k][1].reshape(L, 1)), 1)
        if not torch.equal(got, refs[i]):
            bad += 1
            d = (got - refs[i]).abs(); nz = (d > 0).nonzero()       # atom 5 = the confidence
            key = (tuple(sorted(set((nz[:, 0] // 16 * 16).tolist()))), tuple(sorted(set((nz[:, 1] // 3).tolist()))))
            pat[key] = pat.get(key, 0) + 1
    pipe.close()
print(f"mismatching results: {bad} of {total}; patterns (residue block starts, atoms): {pat}")
