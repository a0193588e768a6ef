#!/usr/bin/env python
"""Build container only: how stable is the reference's recycling at the headline size with the coordinate GRU's MDS
columns at FULL gain?  Runs the CPU oracle (bit-identical to the reference, tests/golden/REPORT.txt)
on bench target 0 with coord_fc fitted to the protein-like trace, at 8 and at 4 threads, and prints the deviation
between the two runs per pass - the thread-count noise a fixture at that depth would carry.

    python tools/explore_fullgain.py --n 10 --m 0 [--ridge 1e-3] [--eps 1.0]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from dmpfold2_amd import synth          # noqa: E402
import dmpfold_oracle as O              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--m", type=int, default=0)
    ap.add_argument("--ridge", type=float, default=1e-3)
    ap.add_argument("--eps", type=float, default=1.0)
    ap.add_argument("--threads", default="8,4")
    a = ap.parse_args()
    import make_goldens as G            # imports the reference too (fit_coord_fc / protein_like_trace live there)
    sd = synth.synth_weights(0, coord_scale=5.0)
    rows = synth.synth_msa(300, 2000, 0)
    if a.eps != 1.0:
        for k in ("coord_gru.weight_ih_l0", "coord_gru.weight_ih_l0_reverse"):
            w = np.array(sd[k]).copy()
            w[:, 512:520] *= np.float32(a.eps)
            sd[k] = w
    sd["coord_fc.weight"] = G.fit_coord_fc(sd, rows, G.protein_like_trace(300, 0), a.ridge)
    W = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    alnmat = O.encode_aln(rows)
    runs = []
    for t in (int(x) for x in a.threads.split(",")):
        torch.set_num_threads(t)
        cap = {}
        coords, confs = O.predict(alnmat, W, None, a.n, a.m, "canonical", cap)
        runs.append((t, coords, confs, cap))
        print(f"threads={t} done: conf means", [round(float(cap[f'p{p}.conf'].mean()), 6) for p in range(a.n + 1)], flush=True)
    t0, c0, f0, cap0 = runs[0]
    for t, c, f, cap in runs[1:]:
        per = [G.rmsd(cap0[f"p{p}.ca"], cap[f"p{p}.ca"]) for p in range(a.n + 1)]
        print(f"n={a.n} m={a.m} ridge={a.ridge} eps={a.eps}: {t0} vs {t} threads: per-pass CA-RMSD",
              " ".join(f"{x:.2e}" for x in per), "| final", f"{G.rmsd(c0[:, 1], c[:, 1]):.2e}",
              "dconf", f"{float((f0 - f).abs().max()):.2e}", flush=True)
    ca = runs[0][1][:, 1]
    b = (ca[1:] - ca[:-1]).norm(dim=1)
    print("final trace: bonds", float(b.min()), float(b.max()), "Rg", float((ca - ca.mean(0)).norm(dim=1).pow(2).mean().sqrt()))


if __name__ == "__main__":
    main()
