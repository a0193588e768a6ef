#!/usr/bin/env python
"""GPU box: the vertical GRU of K targets - K legacy chains side by side on K streams (round 2), K group-kernel
chains side by side, and ONE group chain serving all K (round 3) - plus the bitwise check group == alone.

    python tools/time_vgru_group.py [K=4] [L=300] [N=2000]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dmpfold2_amd import synth, _lib                     # noqa: E402
from dmpfold2_amd.predict import Engine, encode_aln     # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
engs = []
for i in range(K):
    e = Engine(dev, L, N, stream=torch.cuda.Stream(dev))
    e.set_weights(sd)
    engs.append(e)
lib = engs[0].lib
msas = [torch.from_numpy(encode_aln(synth.synth_msa(L, N, 3 + i))).to(dev) for i in range(K)]
outs = [torch.empty(L, 512, device=dev) for _ in range(K)]


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def side_by_side(k):
    def run():
        cur = torch.cuda.current_stream()
        for i in range(k):
            engs[i]._stream.wait_stream(cur)
            _lib.check(lib.dmp_gru_vertical(engs[i].ctx, msas[i].data_ptr(), N, L, outs[i].data_ptr(), engs[i].stream()))
        for i in range(k):
            cur.wait_stream(engs[i]._stream)
    return run


def grouped(k):
    ctxs = (C.c_void_p * k)(*[engs[i].ctx for i in range(k)])
    mp = (C.c_void_p * k)(*[msas[i].data_ptr() for i in range(k)])
    op = (C.c_void_p * k)(*[outs[i].data_ptr() for i in range(k)])
    Ns, Ls = (C.c_int * k)(*([N] * k)), (C.c_int * k)(*([L] * k))

    def run():
        cur = torch.cuda.current_stream()
        engs[0]._stream.wait_stream(cur)
        _lib.check(lib.dmp_gru_vertical_group(ctxs, k, mp, Ns, Ls, op, engs[0].stream()))
        cur.wait_stream(engs[0]._stream)
    return run


for e in engs:
    e.set_option("vgru_legacy", 1)
print(f"L={L} N={N} legacy kernel, chains side by side:",
      "  ".join("%d: %.1f ms" % (k, timed(side_by_side(k))) for k in range(1, K + 1)), flush=True)
for e in engs:
    e.set_option("vgru_legacy", 0)
print("group kernel, chains side by side:           ",
      "  ".join("%d: %.1f ms" % (k, timed(side_by_side(k))) for k in range(1, K + 1)), flush=True)
alone = []
for i in range(K):
    side_by_side(1)()
    _lib.check(lib.dmp_gru_vertical(engs[i].ctx, msas[i].data_ptr(), N, L, outs[i].data_ptr(), engs[i].stream()))
    torch.cuda.synchronize()
    alone.append(outs[i].clone())
print("group kernel, ONE chain for k targets:       ",
      "  ".join("%d: %.1f ms" % (k, timed(grouped(k))) for k in range(1, K + 1)), flush=True)
grouped(K)()
torch.cuda.synchronize()
print("group of %d == each alone, bitwise:" % K, all(torch.equal(outs[i], alone[i]) for i in range(K)),
      " finite:", all(bool(torch.isfinite(o).all()) for o in outs), flush=True)
for nw in ():
    pass
