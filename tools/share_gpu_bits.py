#!/usr/bin/env python
"""GPU box: are a prediction's bits the same while ANOTHER PROCESS keeps the same GPU busy?  (round 4: the two-ranks-on-
one-GPU test of bench.py saw a fresh engine's result differ from the scheduler's.)  The reference bits are computed on
the quiet GPU; then a second process runs predictions in a loop and the first repeats its own with several option sets.

    python tools/share_gpu_bits.py            # parent
"""
import os
import subprocess
import sys
import time

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
msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, 0))).to(dev)


def engine(**opts):
    e = Engine(dev, L, N)
    e.set_weights(sd)
    for k, v in opts.items():
        e.set_option(k, v)
    return e


if len(sys.argv) > 1 and sys.argv[1] == "load":
    e = engine(vgru_persistent=0)
    t_end = time.time() + float(sys.argv[2])
    while time.time() < t_end:
        e.predict_device(msa, None, 2, 10)
        e.sync_check()
    sys.exit(0)

variants = {
    "per-row, cluster tridiag": dict(vgru_persistent=0),
    "per-row, per-step tridiag": dict(vgru_persistent=0, tridiag_cluster=0),
    "per-row, agent-scope hand-offs": dict(vgru_persistent=0, cluster_local=0),
    "per-row, single-workgroup tridiag + refine": dict(vgru_persistent=0, tridiag_single=1, refine_single=1),
}
engines = {name: engine(**o) for name, o in variants.items()}
ref = {}
for name, e in engines.items():
    c, f = e.predict_device(msa, None, 3, 20)
    e.sync_check()
    ref[name] = (c.clone(), f.clone())
base = ref["per-row, cluster tridiag"]
for name in variants:
    print(f"quiet GPU: {name:45s} == first variant: {bool(torch.equal(ref[name][0], base[0]))}", flush=True)
load = subprocess.Popen([sys.executable, os.path.abspath(__file__), "load", "60"])
time.sleep(15)
for rep in range(3):
    for name, e in engines.items():
        try:
            c, f = e.predict_device(msa, None, 3, 20)
            e.sync_check()
            same = bool(torch.equal(c, ref[name][0]) and torch.equal(f, ref[name][1]))
            d = float((c - ref[name][0]).abs().max())
            print(f"beside another process, run {rep}: {name:45s} same bits: {same}  max|dcoords| {d:.3e}", flush=True)
        except Exception as ex:                         # a fault is a result too
            print(f"beside another process, run {rep}: {name:45s} {type(ex).__name__}: {str(ex)[:120]}", flush=True)
load.wait()
