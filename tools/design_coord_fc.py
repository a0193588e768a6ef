#!/usr/bin/env python
"""GPU box: design a coord_fc (3 x 512) for which the FIRST-pass CA trace of a synthetic alignment is
protein-like (3.8 A bonds, no clashes) with weights as small as that allows, and measure how stable the
benchmark's full setting (iterations=10, minsteps=100) is on it.

The first trace is G W^T with G (L x 512) the pass-0 coordinate-GRU output, independent of coord_fc
(tests/golden/make_goldens.fit_coord_fc).  Instead of regressing onto a foreign structure - which at L=300
needs weights large enough to make recycling expansive: the reference's own 8- and 4-thread runs then differ by
tens of Angstrom - W minimises the minimiser's own energy of G W^T (bonds, repulsion) plus a weight penalty.
Stability proxy: the HIP path's three convolution arithmetics (f16x3, exact f32, bf16x6) against each other
on every pass; a fixture on which they agree to a few 1e-4 A is one on which the reference's thread-count
noise is of that size too.  The chosen W goes to gpurun_out/ and from there into the golden generator.
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
ap.add_argument("--decay", type=float, nargs="*", default=[1e-3, 1e-2, 3e-2])
ap.add_argument("--init", type=float, default=2.0)
ap.add_argument("--out", default="gpurun_out/coord_fc")
a = ap.parse_args()
dev = torch.device("cuda:0")
sd = synth.synth_weights(0, coord_scale=5.0)
st = Stages(sd, a.L, a.N)
eng = st.eng
L = a.L
alnmat = encode_aln(synth.synth_msa(a.L, a.N, a.seed))
eng.predict(alnmat, None, 0, 0)
eng.sync_check()
mat1d = eng.fetch("mat1d", 512 * L).reshape(512, L).clone()
mds = eng.fetch("mds", L * 8).reshape(L, 8).clone()
emb = torch.cat((mat1d.t().contiguous(), mds), dim=1).contiguous()
G = st.gru_bidir(1, emb).clone()                      # (L, 512)
torch.cuda.synchronize()
print("G: rows", G.shape, "row norm", float(G.norm(dim=1).mean()), flush=True)


def energy(x):
    d = torch.cdist(x, x) + torch.eye(L, device=dev) * 1e3
    i = torch.arange(L - 1, device=dev)
    bond = ((d[i, i + 1] - 3.8) ** 2).sum()
    i2 = torch.arange(L - 2, device=dev)
    ang = (torch.relu(5.2 - d[i2, i2 + 2]) ** 2).sum() + (torch.relu(d[i2, i2 + 2] - 7.0) ** 2).sum()
    mask = (torch.arange(L, device=dev)[:, None] - torch.arange(L, device=dev)[None, :]).abs() > 2
    rep = (torch.relu(4.6 - d)[mask] ** 2).sum() / 2
    r = (x - x.mean(0)).norm(dim=1)
    comp = (torch.relu(r - (3.3 * L ** (1 / 3) + 6.0)) ** 2).sum()
    return bond + ang + rep + 0.1 * comp


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
for decay in a.decay:
    gen = torch.Generator(device="cpu").manual_seed(1)
    W = ((torch.rand(3, 512, generator=gen) * 2 - 1) * a.init).to(dev).requires_grad_(True)
    opt = torch.optim.Adam([W], lr=0.02)
    for it in range(6000):
        opt.zero_grad()
        x = G @ W.t()
        loss = energy(x) + decay * L * (W ** 2).sum()
        loss.backward()
        opt.step()
    Wf = W.detach().clone()
    x = G @ Wf.t()
    print(f"decay {decay:g}: energy {float(energy(x)):.3f} max|W| {float(Wf.abs().max()):.2f} rms W "
          f"{float((Wf ** 2).mean().sqrt()):.2f}  {stats(x)}", flush=True)
    sd2 = dict(sd)
    sd2["coord_fc.weight"] = Wf.cpu().numpy()
    eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()})
    runs = {}
    for mode in (0, 1, 2):
        eng.set_option("conv_mode", mode)
        coords, confs = eng.predict(alnmat, None, 10, 100)
        eng.sync_check()
        runs[mode] = (coords.cpu(), confs.cpu(), eng.fetch("ca_pass", 11 * L * 3).reshape(11, L, 3).cpu(),
                      eng.fetch("conf_means", 11).cpu())
    eng.set_option("conv_mode", 0)
    for m in (1, 2):
        per = [rmsd(runs[0][2][p], runs[m][2][p]) for p in range(11)]
        print(f"   mode 0 vs {m}: final CA-RMSD {rmsd(runs[0][0][:, 1], runs[m][0][:, 1]):.2e} max|dconf| "
              f"{float((runs[0][1] - runs[m][1]).abs().max()):.2e} per pass " + " ".join(f"{v:.1e}" for v in per), flush=True)
    print("   conf means", " ".join(f"{float(v):.4f}" for v in runs[0][3]),
          " refined first trace:", stats(runs[0][2][0].to(dev)), flush=True)
    np.save(f"{a.out}_L{a.L}_decay{decay:g}.npy", Wf.cpu().numpy())
