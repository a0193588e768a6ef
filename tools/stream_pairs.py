#!/usr/bin/env python
"""GPU box: which HIP streams get in each other's way?  Two vertical-GRU chains (dependent kernel chains of
2000 launches) run on every pair of NS streams; pairs that share a hardware pipe serialise."""
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
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
engs = []
for i in range(4):
    e = Engine(dev, L, N, stream=streams[i])
    e.set_weights(sd)
    engs.append(e)
msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, 3))).to(dev)
outs = [torch.empty(L, 512, device=dev) for _ in range(4)]


def run(ids, reps=2):
    best = 1e9
    for i, s in enumerate(ids):
        engs[i]._stream = streams[s]
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(len(ids)):
            engs[i]._stream.wait_stream(torch.cuda.current_stream())
            engs[i].lib.dmp_gru_vertical(engs[i].ctx, msa.data_ptr(), N, L, outs[i].data_ptr(), engs[i].stream())
        for i in range(len(ids)):
            torch.cuda.current_stream().wait_stream(engs[i]._stream)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


run([0, 1])
print("GPU_MAX_HW_QUEUES=%s, %d streams; ms for two chains on streams (row, column):" % (os.environ.get("GPU_MAX_HW_QUEUES"), NS))
for i in range(NS):
    print("%2d: " % i + " ".join("%5.1f" % run([i, j], 1) if j > i else "     " for j in range(NS)), flush=True)
for quad in ([0, 1, 2, 3], [0, 2, 4, 6], [1, 3, 5, 7], [0, 1, 4, 5], [4, 5, 6, 7], [2, 3, 6, 7]):
    if max(quad) < NS:
        print("four chains on streams", quad, "%.1f ms" % run(quad), flush=True)
