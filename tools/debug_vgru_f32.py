#!/usr/bin/env python
"""GPU box: the float32 vertical GRU (vgru_f32.hip) against a float64 NumPy recurrence, layer by layer after N = 1, 2, 3
rows - where (hidden row, column) does it differ?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dmpfold2_amd import synth                       # noqa: E402
from dmpfold2_amd.predict import encode_aln          # noqa: E402
from abi import Stages                               # noqa: E402

sd = synth.synth_weights(0, coord_scale=5.0)
st = Stages(sd, 64, 16)
eng = st.eng
eng.set_option("vgru_f32", 1)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items() if k.startswith("vgru.")}


def sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def cell(x, h, l):
    gi = x @ W[f"vgru.weight_ih_l{l}"].T + W[f"vgru.bias_ih_l{l}"]
    gh = h @ W[f"vgru.weight_hh_l{l}"].T + W[f"vgru.bias_hh_l{l}"]
    r = sig(gi[:, :512] + gh[:, :512])
    z = sig(gi[:, 512:1024] + gh[:, 512:1024])
    n = np.tanh(gi[:, 1024:] + r * gh[:, 1024:])
    return (h - n) * z + n


for persistent in (1, 0):
    eng.set_option("vgru_persistent", persistent)
    for N in (1, 2, 3):
        m = encode_aln(synth.synth_msa(L, N, 5))
        out = st.gru_vertical(m).cpu().numpy()
        Lb = (L + 31) // 32 * 32
        g0 = eng.fetch("vgru_h0", 512 * Lb).cpu().numpy().reshape(128, Lb, 4).transpose(0, 2, 1).reshape(512, Lb)[:, :L]
        g1 = eng.fetch("vgru_h1", 512 * Lb).cpu().numpy().reshape(128, Lb, 4).transpose(0, 2, 1).reshape(512, Lb)[:, :L]
        h0 = np.zeros((L, 512)); h1 = np.zeros((L, 512))
        for t in range(N):
            h0 = cell(np.eye(22)[m[t]], h0, 0)
            h1 = cell(h0, h1, 1)
        print(f"persistent={persistent} N={N} faults {eng.sync_faults()}  out vs h1: {np.abs(out - h1).max():.2e}")
        for name, got, ref in (("h0", g0, h0.T), ("h1", g1, h1.T)):
            d = np.abs(got - ref)
            bad_r = np.where(d.max(axis=1) > 1e-5)[0]
            bad_c = np.where(d.max(axis=0) > 1e-5)[0]
            print(f"   {name}: max|d| {d.max():.3e}  bad rows {len(bad_r)}/512 (mod 16 hist {np.bincount(bad_r % 16, minlength=16).tolist()}) "
                  f"bad cols {bad_c.tolist()[:40]}")
            if len(bad_r):
                r = bad_r[0]
                print(f"      row {r}: got {np.array2string(got[r, :6], precision=5)} ref {np.array2string(ref[r, :6], precision=5)}")
