#!/usr/bin/env python
"""GPU box: single-stream prediction latency with the cluster minimiser and with the single-workgroup one."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dmpfold2_amd import synth
from dmpfold2_amd.predict import Engine, encode_aln
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
for L, N, it, ms in ((200, 1000, 10, 100), (300, 2000, 10, 100), (500, 3000, 30, 200)):
    e = Engine(dev, L, N); e.set_weights(sd)
    a = encode_aln(synth.synth_msa(L, N, seed=1))
    for single in (0, 1):
        e.set_option("refine_single", single)
        e.predict(a, None, it, ms); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2): e.predict(a, None, it, ms)
        torch.cuda.synchronize()
        print(f"L={L} N={N} {it}+{ms}: refine_single={single}: {(time.perf_counter() - t0) / 2 * 1e3:.1f} ms")
    e.close()
