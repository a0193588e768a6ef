#!/usr/bin/env python
"""GPU box: the scheduler's results against a fresh engine's, target by target, for S engines and both forms of the
vertical GRU (round 4: `bench.py --streams 2 --vgru-per-row` reported a mismatch).   python tools/scheduler_vs_engine_bits.py [S]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dmpfold2_amd import synth                          # noqa: E402
from dmpfold2_amd.predict import Engine, Pipeline, encode_aln     # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L, N = 300, 2000
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
msas = [torch.from_numpy(encode_aln(synth.synth_msa(L, N, i))).to(dev) for i in range(4 * S)]
for persistent in (0, 1):
    pipe = Pipeline(dev, L, N, sd, streams=S)
    for e in pipe.engines:
        e.set_option("vgru_persistent", persistent)
    tickets = [pipe.submit(m, 10, 100) for m in msas]
    pipe.drain()
    pipe.sync_check()
    res = [pipe.result(t) for t in tickets]
    single = Engine(dev, L, N)
    single.set_option("vgru_persistent", persistent)
    single.share_weights(pipe.engines[0])
    for i, m in enumerate(msas):
        c, f = single.predict_device(m, None, 10, 100)
        single.sync_check()
        c0, f0 = pipe.engines[0].predict_device(m, None, 10, 100)
        pipe.engines[0].sync_check()
        print(f"S={S} persistent={persistent} target {i}: scheduler == fresh engine {bool(torch.equal(c, res[i][0]))} "
              f"(max|d| {float((c - res[i][0]).abs().max()):.2e}), scheduler == its engine 0 alone {bool(torch.equal(c0, res[i][0]))}, "
              f"fresh == engine 0 alone {bool(torch.equal(c, c0))}", flush=True)
    single.close()
    pipe.close()
