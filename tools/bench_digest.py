#!/usr/bin/env python
"""GPU box: the digest bench.py stores in profiles/bench_digest.json (bench target 0 - alignment seed 0, L=300, N=2000,
10 iterations + 100 minimiser steps - alone on one engine: SHA-256 of coords + confs), without the bench around it.
Run after a change of arithmetic, copy the line into profiles/bench_digest.json."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmpfold2_amd import synth                        # noqa: E402
from dmpfold2_amd.predict import Engine, encode_aln   # noqa: E402

L, N, ITERS, MINSTEPS = 300, 2000, 10, 100
eng = Engine("cuda:0", L, N)
eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()})
msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, seed=0))).to("cuda:0")
c, f = eng.predict_device(msa, None, ITERS, MINSTEPS)
eng.sync_check()
print(json.dumps({f"L{L}_N{N}_n{ITERS}_m{MINSTEPS}_seed0": hashlib.sha256(c.cpu().numpy().tobytes() + f.cpu().numpy().tobytes()).hexdigest()}))
