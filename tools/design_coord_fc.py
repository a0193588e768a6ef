#!/usr/bin/env python
"""GPU box: design a coord_fc (3 x 512) for which the FIRST-pass CA trace of a synthetic alignment is
protein-like (3.8 A bonds, no clashes) with weights as small as that allows, and measure how stable the
benchmark's full setting (iterations=10, minsteps=100) is on it.

The first trace is G W^T with G (L x 512) the pass-0 coordinate-GRU output, independent of coord_fc
(tests/golden/make_goldens.fit_coord_fc).  Regressing onto a foreign structure with a small ridge needs weights
large enough at L=300 to make recycling expansive (the reference's own 8- and 4-thread runs then differ by tens
of Angstrom: fixture fitns_L300_N2000_n10_m100 of the first attempt).  Designs compared here: (a) the principal
axes of G itself scaled to 3.8 A mean bonds (the smallest weights for a given extent), (b) ridge regression onto
the stored protein-like target with increasing ridge.  Stability proxy: the HIP path's three convolution
arithmetics (f16x3, exact f32, bf16x6) against each other on every pass, at n=10 with m=0 and m=100; a fixture
on which they agree to a few 1e-4 A is one on which the reference's thread-count noise is of that size too.
The chosen W goes to gpurun_out/ and from there into the golden generator.
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
ap.add_argument("--ridge", type=float, nargs="*", default=[0.1, 1.0, 10.0, 100.0])
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
Gd = G.double()
Gc = Gd - Gd.mean(0, keepdim=True)
sv = torch.linalg.svdvals(Gc)
print("singular values of centred G:", " ".join(f"{float(v):.3g}" for v in sv[:12]), "...", f"{float(sv[-1]):.3g}", flush=True)
designs = []
# (a) principal axes of G itself: the smallest weights that give the trace a given extent
U, S, Vh = torch.linalg.svd(Gc, full_matrices=False)
for k0 in (0, 1):
    Wp = Vh[k0:k0 + 3]                                 # (3, 512)
    x = Gd @ Wp.t()
    bond = float((x[1:] - x[:-1]).norm(dim=1).mean())
    designs.append((f"pca{k0}", (Wp * (3.8 / bond)).float()))
# (b) ridge regression onto the stored protein-like target, increasing ridge = smaller weights
gpath = os.path.join(ROOT, "tests", "golden", "fitns_L300_N2000_n10_m100.npz")
if os.path.exists(gpath) and a.L == 300:
    T = torch.from_numpy(np.load(gpath)["target_ca"]).double().to(dev)
    T = T - T.mean(0, keepdim=True)
    for ridge in a.ridge:
        A = Gd @ Gd.t() + ridge * torch.eye(L, dtype=torch.float64, device=dev)
        designs.append((f"ridge{ridge:g}", (Gd.t() @ torch.linalg.solve(A, T)).t().float()))
for tag, Wf in designs:
    Wf = Wf.contiguous()
    x = G @ Wf.t()
    print(f"{tag}: max|W| {float(Wf.abs().max()):.2f} rms W {float((Wf ** 2).mean().sqrt()):.3f}  first trace: {stats(x)}", flush=True)
    sd2 = dict(sd)
    sd2["coord_fc.weight"] = Wf.cpu().numpy()
    eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()})
    for (n, m) in ((10, 0), (10, 100)):
        runs = {}
        for mode in (0, 1, 2):
            eng.set_option("conv_mode", mode)
            coords, confs = eng.predict(alnmat, None, n, m)
            eng.sync_check()
            runs[mode] = (coords.cpu(), confs.cpu(), eng.fetch("ca_pass", 11 * L * 3).reshape(11, L, 3).cpu(),
                          eng.fetch("conf_means", 11).cpu())
        eng.set_option("conv_mode", 0)
        for mo in (1, 2):
            per = [rmsd(runs[0][2][p], runs[mo][2][p]) for p in range(11)]
            print(f"   n={n} m={m} mode 0 vs {mo}: final CA-RMSD {rmsd(runs[0][0][:, 1], runs[mo][0][:, 1]):.2e} max|dconf| "
                  f"{float((runs[0][1] - runs[mo][1]).abs().max()):.2e} per pass " + " ".join(f"{v:.1e}" for v in per), flush=True)
        print("   conf means", " ".join(f"{float(v):.4f}" for v in runs[0][3]),
              " first trace of the run:", stats(runs[0][2][0].to(dev)), " final:", stats(runs[0][0][:, 1].to(dev)), flush=True)
    np.save(f"{a.out}_L{a.L}_{tag}.npy", Wf.cpu().numpy())
