#!/usr/bin/env python
"""GPU box: K contexts running dmp_gru_vertical at the same time on their own streams (the front-end hole
of the throughput scheduler is four of them).  GPU_MAX_HW_QUEUES from the environment."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dmpfold2_amd import synth                          # noqa: E402
from dmpfold2_amd.predict import Engine, encode_aln     # noqa: E402

L, N = 300, 2000
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
engs = []
for i in range(K):
    e = Engine(dev, L, N, stream=torch.cuda.Stream(dev))
    e.set_weights(sd)
    engs.append(e)
msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, 3))).to(dev)
outs = [torch.empty(L, 512, device=dev) for _ in range(K)]


def run(k, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(k):
            engs[i]._stream.wait_stream(torch.cuda.current_stream())
            engs[i].lib.dmp_gru_vertical(engs[i].ctx, msa.data_ptr(), N, L, outs[i].data_ptr(), engs[i].stream())
        for i in range(k):
            torch.cuda.current_stream().wait_stream(engs[i]._stream)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print("GPU_MAX_HW_QUEUES=%s:" % os.environ.get("GPU_MAX_HW_QUEUES"),
      "  ".join("%d chains %.1f ms" % (k, run(k)) for k in range(1, K + 1)), flush=True)
