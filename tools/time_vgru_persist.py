#!/usr/bin/env python
"""GPU box: the vertical GRU as ONE persistent weight-stationary launch (round 4, option vgru_persistent = 1) against the
launch chain of rounds 2-3 (= 0): results (max |difference|, grouping bit-identity, a small case against the oracle's
nn.GRU) and the time of one chain serving k = 1 .. K targets.

    [VGRU_F32=0|1|2] python tools/time_vgru_persist.py [K=8] [L=300] [N=2000]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dmpfold2_amd import synth, _lib                     # noqa: E402
from dmpfold2_amd.predict import Engine, encode_aln     # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
lead = Engine(dev, L, N, stream=torch.cuda.Stream(dev))
lead.set_weights(sd)
lib = lead.lib
# VGRU_F32: 0 the split-f16 form, 1 float32 MFMAs (vgru_f32.hip), 2 three bf16 pieces per operand (vgru_x3.hip)
lead.set_option("vgru_f32", int(os.environ.get("VGRU_F32", "0")))
print("persistent form available:", lead.get_option("vgru_persistent"), " float32:", lead.get_option("vgru_f32"), flush=True)


def chain(msas, persistent):
    k = len(msas)
    lead.set_option("vgru_persistent", persistent)
    outs = [torch.empty(m.shape[1], 512, device=dev) for m in msas]
    ctxs = (C.c_void_p * k)(*[lead.ctx] * k)
    mp = (C.c_void_p * k)(*[m.data_ptr() for m in msas])
    op = (C.c_void_p * k)(*[o.data_ptr() for o in outs])
    Ns = (C.c_int * k)(*[m.shape[0] for m in msas])
    Ls = (C.c_int * k)(*[m.shape[1] for m in msas])

    def run():
        cur = torch.cuda.current_stream()
        lead._stream.wait_stream(cur)
        _lib.check(lib.dmp_gru_vertical_group(ctxs, k, mp, Ns, Ls, op, lead.stream()))
        cur.wait_stream(lead._stream)
    return run, outs


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


# ---- results: a small ragged group against the oracle and the launch chain
import dmpfold_oracle as O                               # noqa: E402  (checker)
shapes = [(82, 100), (33, 7), (128, 64), (40, 1), (50, 257)]
small = [torch.from_numpy(encode_aln(synth.synth_msa(l, n, 70 + i))).to(dev) for i, (l, n) in enumerate(shapes)]
if L < 128 or N < 257:
    small = [m for m in small if m.shape[1] <= L and m.shape[0] <= N]
run_p, out_p = chain(small, 1)
run_p()
torch.cuda.synchronize()
print("faults after the persistent chain:", lead.sync_faults(), flush=True)
run_c, out_c = chain(small, 0)
run_c()
torch.cuda.synchronize()
W = {k: v for k, v in sd.items()}
for m, a, b in zip(small, out_p, out_c):
    x = W["embed.weight"][m.cpu().long()]
    ref = O._gru(W, "vgru", x, 22, 512, 2, False, False)[-1]
    print(f"  L={m.shape[1]:4d} N={m.shape[0]:4d}: persistent vs oracle {float((a.cpu() - ref).abs().max()):.2e}, launch chain vs oracle "
          f"{float((b.cpu() - ref).abs().max()):.2e}, persistent vs launch chain {float((a - b).abs().max()):.2e}", flush=True)
# grouping does not change a member's bits
for i, m in enumerate(small):
    r1, o1 = chain([m], 1)
    r1()
    torch.cuda.synchronize()
    print(f"  member {i} alone == in the group (persistent): {bool(torch.equal(o1[0], out_p[i]))}", flush=True)

# ---- time: ONE chain for k targets
msas = [torch.from_numpy(encode_aln(synth.synth_msa(L, N, 3 + i))).to(dev) for i in range(K)]
for persistent in (0, 1):
    times = []
    for k in range(1, K + 1):
        run, outs = chain(msas[:k], persistent)
        run()
        times.append((k, timed(run)))
    print(f"L={L} N={N} vgru_persistent={persistent}: one chain for k targets:",
          "  ".join("%d: %.1f ms (%.1f us/row)" % (k, t, t * 1e3 / (N + 1)) for k, t in times), flush=True)
# at full size too a member's bits do not depend on the chain it ran in, in either form
for persistent in (1, 0):
    ref0 = None
    verdict = []
    for k in range(1, K + 1):
        run, outs = chain(msas[:k], persistent)
        run()
        torch.cuda.synchronize()
        if ref0 is None:
            ref0 = outs[0].clone()
        verdict.append(bool(torch.equal(outs[0], ref0)))
    print(f"L={L} N={N} vgru_persistent={persistent}: target 0 in a chain of k = 1..{K} has the bits of k = 1:", verdict, flush=True)
print("faults:", lead.sync_faults())
run_p, out_p = chain(msas[:K], 1)
run_p()
run_c, out_c = chain(msas[:K], 0)
run_c()
torch.cuda.synchronize()
print("full size, persistent vs launch chain: max |d| =", max(float((a - b).abs().max()) for a, b in zip(out_p, out_c)))
