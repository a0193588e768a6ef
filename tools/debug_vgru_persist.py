#!/usr/bin/env python
"""GPU box: where does the persistent vertical GRU differ from the launch chain?  Layer-0 / layer-1 state after
N = 1, 2, 3 rows, by hidden row and column."""
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

st = Stages(synth.synth_weights(0, coord_scale=5.0), 64, 16)
eng = st.eng
L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for N in (1, 2):
    m = encode_aln(synth.synth_msa(L, N, 5))
    res = {}
    for mode in (0, 1):
        eng.set_option("vgru_persistent", mode)
        out = st.gru_vertical(m).cpu().numpy()
        Lb = (L + 31) // 32 * 32
        h0 = eng.fetch("vgru_h0", 512 * Lb).cpu().numpy().reshape(128, Lb, 4).transpose(0, 2, 1).reshape(512, Lb)[:, :L]
        h1 = eng.fetch("vgru_h1", 512 * Lb).cpu().numpy().reshape(128, Lb, 4).transpose(0, 2, 1).reshape(512, Lb)[:, :L]
        res[mode] = (out, h0, h1)
    print("faults", eng.sync_faults())
    for name, k in (("h0", 1), ("h1", 2)):
        a, b = res[1][k], res[0][k]
        d = np.abs(a - b)
        print(f"N={N} {name}: max|d| {d.max():.3e}; rows with error > 1e-5: {int((d.max(axis=1) > 1e-5).sum())}/512; "
              f"cols: {int((d.max(axis=0) > 1e-5).sum())}/{L}")
        bad_rows = np.where(d.max(axis=1) > 1e-5)[0]
        print("   bad rows mod 16 histogram:", np.bincount(bad_rows % 16, minlength=16).tolist())
        print("   bad rows // 16 (CU) first 40:", sorted(set((bad_rows // 16).tolist()))[:40])
        bad_cols = np.where(d.max(axis=0) > 1e-5)[0]
        print("   bad cols:", bad_cols.tolist()[:48])
        print("   persistent rows 0..15 col 0:", np.array2string(a[:16, 0], precision=5))
        print("   chain      rows 0..15 col 0:", np.array2string(b[:16, 0], precision=5))
        # is a persistent row some OTHER row of the chain (a permutation inside the CU's 16 rows)?
        perm = [int(np.argmin(np.abs(b[:16, :] - a[r:r + 1, :]).max(axis=1))) for r in range(16)]
        err = [float(np.abs(b[perm[r], :] - a[r, :]).max()) for r in range(16)]
        print("   best-matching chain row for persistent rows 0..15:", perm, "max err", max(err))

# hypothesis test: at row t = 1 the finishing threads used hp[row 4g] for all four rows of their group
sd = synth.synth_weights(0, coord_scale=5.0)
m = encode_aln(synth.synth_msa(L, 2, 5))
eng.set_option("vgru_persistent", 0)
st.gru_vertical(m[:1])
Lb = (L + 31) // 32 * 32
s1 = eng.fetch("vgru_h0", 512 * Lb).cpu().numpy().reshape(128, Lb, 4).transpose(0, 2, 1).reshape(512, Lb)[:, :L]
st.gru_vertical(m)
s2 = eng.fetch("vgru_h0", 512 * Lb).cpu().numpy().reshape(128, Lb, 4).transpose(0, 2, 1).reshape(512, Lb)[:, :L]
eng.set_option("vgru_persistent", 1)
st.gru_vertical(m)
p2 = eng.fetch("vgru_h0", 512 * Lb).cpu().numpy().reshape(128, Lb, 4).transpose(0, 2, 1).reshape(512, Lb)[:, :L]
Wi, Wh = sd["vgru.weight_ih_l0"], sd["vgru.weight_hh_l0"]
bi, bh = sd["vgru.bias_ih_l0"], sd["vgru.bias_hh_l0"]
x = np.eye(22, dtype=np.float32)[m[1]]                      # (L, 22)
gi = x @ Wi.T + bi
gh = s1.T @ Wh.T + bh
r = 1 / (1 + np.exp(-(gi[:, :512] + gh[:, :512])))
z = 1 / (1 + np.exp(-(gi[:, 512:1024] + gh[:, 512:1024])))
n = np.tanh(gi[:, 1024:] + r * gh[:, 1024:])
ref = ((s1.T - n) * z + n).T
print("numpy layer 0 at row 1 vs launch chain:", np.abs(ref - s2).max())
hp_b = np.repeat(s1.reshape(128, 4, L)[:, :1, :], 4, axis=1).reshape(512, L)      # hp of row 4g for the whole group
alt = ((hp_b.T - n) * z + n).T
print("numpy with hp broadcast from row 4g vs persistent:", np.abs(alt - p2).max(), " (plain numpy vs persistent:", np.abs(ref - p2).max(), ")")
for k in (1, 2, 3):
    hp_k = np.repeat(s1.reshape(128, 4, L)[:, k:k + 1, :], 4, axis=1).reshape(512, L)
    print(f"  hp broadcast from row 4g+{k}:", np.abs((((hp_k.T - n) * z + n).T) - p2).max())
