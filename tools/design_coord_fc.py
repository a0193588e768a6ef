#!/usr/bin/env python
"""GPU box: design synthetic weights on which the benchmark's FULL setting (iterations=10, minsteps=100) at
L=300, N=2000 is reference-stable, so that the reference itself can pin it (tests/golden/make_goldens.py).

The first CA trace is G W^T with G (L x 512) the pass-0 coordinate-GRU output, independent of coord_fc.  What was
tried and measured (gpurun_out r03a / r03c, stability proxy = the HIP path's three convolution arithmetics against
each other on every pass, at n=10 with m=0 and m=100):
  * ridge regression of coord_fc onto a protein-like 300-residue trace, ridge 0.1 (the fixture of the first
    attempt): protein-like after refinement, but rms|W| = 7.7 makes RECYCLING expansive even at m=0 (3e-4 A at
    pass 0 -> 0.4 A at pass 10; the reference's own 8- and 4-thread runs end 120 A apart);
  * larger ridges (1, 10, 100) and the principal axes of G: stable at m=0 (1e-5 .. 3e-4 A) but the first trace is
    collapsed (bonds 0.1-1.2 A) or stretched (bonds up to 96 A) and the minimiser amplifies 1e-4 to 1-3 A;
  * minimising the minimiser's own energy over W (Adam): does not reach a protein-like trace.
The loop gain of recycling is (sensitivity of the coordinate GRU to its 8 MDS inputs) x |W|.  This version lowers the
first factor instead of the second: the 8 MDS columns of coord_gru.weight_ih_l0(_reverse) are scaled by `eps`, and
coord_fc is fitted with a SMALL ridge, so that every pass's trace stays close to the protein-like target (regular
regime of the minimiser: 3.8 A bonds, no clashes) while the trunk, the best-of selection and both refinements run
at the benchmark's size and depth.
"""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dmpfold2_amd import synth
from dmpfold2_amd.predict import encode_aln
from abi import Stages

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=300)
ap.add_argument("--N", type=int, default=2000)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--eps", type=float, nargs="*", default=[0.0, 0.02, 0.1, 0.3, 1.0])
ap.add_argument("--ridge", type=float, nargs="*", default=[1e-3, 1e-1])
ap.add_argument("--out", default="gpurun_out/coord_fc")
ap.add_argument("--n", type=int, default=10, help="iterations of the stability runs")
ap.add_argument("--m", type=int, default=100, help="minimiser steps of the stability runs")
ap.add_argument("--trace-seed", type=int, default=-1, help="protein_like_trace seed (default: the L = 300 fixture's target)")
a = ap.parse_args()
dev = torch.device("cuda:0")
sd = synth.synth_weights(0, coord_scale=5.0)
st = Stages(sd, a.L, a.N)
eng = st.eng
L = a.L
alnmat = encode_aln(synth.synth_msa(a.L, a.N, a.seed))
if a.trace_seed >= 0:
    T = torch.from_numpy(synth.protein_like_trace(a.L, a.trace_seed)).double().to(dev)
else:
    gpath = os.path.join(ROOT, "tests", "golden", "fitns_L300_N2000_n10_m100.npz")
    T = torch.from_numpy(np.load(gpath)["target_ca"]).double().to(dev)
T = T - T.mean(0, keepdim=True)


def stats(x):
    d = torch.cdist(x, x)
    i = torch.arange(L - 1, device=dev)
    b = d[i, i + 1]
    mask = (torch.arange(L, device=dev)[:, None] - torch.arange(L, device=dev)[None, :]).abs() > 1
    return (f"bonds {float(b.min()):.2f}/{float(b.mean()):.2f}/{float(b.max()):.2f} min nonbonded "
            f"{float(d[mask].min()):.2f} pairs<3A {int((d[mask] < 3.0).sum()) // 2} Rg "
            f"{float(((x - x.mean(0)) ** 2).sum(1).mean().sqrt()):.1f}")


def rmsd(p, q):
    return float(((p - q) ** 2).sum(-1).mean().sqrt())


os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
for eps in a.eps:
    sde = dict(sd)
    for k in ("coord_gru.weight_ih_l0", "coord_gru.weight_ih_l0_reverse"):
        w = np.array(sd[k]).copy()
        w[:, 512:520] *= np.float32(eps)
        sde[k] = w
    eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sde.items()})
    eng.predict(alnmat, None, 0, 0)
    eng.sync_check()
    mat1d = eng.fetch("mat1d", 512 * L).reshape(512, L).clone()
    mds = eng.fetch("mds", L * 8).reshape(L, 8).clone()
    emb = torch.cat((mat1d.t().contiguous(), mds), dim=1).contiguous()
    Gd = st.gru_bidir(1, emb).clone().double()           # (L, 512)
    torch.cuda.synchronize()
    for ridge in a.ridge:
        A = Gd @ Gd.t() + ridge * torch.eye(L, dtype=torch.float64, device=dev)
        Wf = (Gd.t() @ torch.linalg.solve(A, T)).t().float().contiguous()
        x = Gd.float() @ Wf.t()
        tag = f"eps{eps:g}_ridge{ridge:g}"
        print(f"{tag}: max|W| {float(Wf.abs().max()):.2f} rms W {float((Wf ** 2).mean().sqrt()):.3f}  first trace: {stats(x)}  "
              f"fit rmsd {rmsd(x.double(), T):.3f}", flush=True)
        sd2 = dict(sde)
        sd2["coord_fc.weight"] = Wf.cpu().numpy()
        eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()})
        for (n, m) in ((a.n, 0), (a.n, a.m)):
            runs = {}
            P = n + 1
            for mode in (0, 1, 2):
                eng.set_option("precision", mode)
                coords, confs = eng.predict(alnmat, None, n, m)
                eng.sync_check()
                runs[mode] = (coords.cpu(), confs.cpu(), eng.fetch("ca_pass", P * L * 3).reshape(P, L, 3).cpu(),
                              eng.fetch("conf_means", P).cpu())
            eng.set_option("precision", 0)
            for mo in (1, 2):
                per = [rmsd(runs[0][2][p], runs[mo][2][p]) for p in range(P)]
                print(f"   n={n} m={m} mode 0 vs {mo}: final CA-RMSD {rmsd(runs[0][0][:, 1], runs[mo][0][:, 1]):.2e} max|dconf| "
                      f"{float((runs[0][1] - runs[mo][1]).abs().max()):.2e} per pass " + " ".join(f"{v:.1e}" for v in per), flush=True)
            moved = [rmsd(runs[0][2][p], runs[0][2][0]) for p in range(P)]
            print("   conf means", " ".join(f"{float(v):.4f}" for v in runs[0][3]),
                  " trace of pass p against pass 0:", " ".join(f"{v:.2g}" for v in moved),
                  " final:", stats(runs[0][0][:, 1].to(dev)), flush=True)
        np.save(f"{a.out}_L{a.L}_{tag}.npy", Wf.cpu().numpy())
