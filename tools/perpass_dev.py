#!/usr/bin/env python
"""GPU box: per-pass CA-RMSD of the HIP path against a reference golden, per convolution mode
(developer diagnostic for error growth through the recycling loop)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmpfold2_amd import synth                      # noqa: E402
from dmpfold2_amd.predict import Engine             # noqa: E402


def rmsd(a, b):
    return float(np.sqrt(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).sum(-1).mean()))


for name in sys.argv[1:] or ["pf10963_n10_m0"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    sd = synth.synth_weights(0, coord_scale=5.0)
    if "coord_gru_mds_scale" in g.files:
        sd = synth.headline_fixture_weights(g["coord_fc"], float(g["coord_gru_mds_scale"]))
    elif "coord_fc" in g.files:
        sd["coord_fc.weight"] = g["coord_fc"]
    n, m = int(g["iterations"]), int(g["minsteps"])
    L = g["coords"].shape[0]
    if "alnmat" in g.files:
        alnmat = g["alnmat"]
    else:                                   # large synthetic alignments are regenerated from their seed
        from dmpfold2_amd.predict import encode_aln
        alnmat = encode_aln(synth.synth_msa(L, int(g["msa_rows"]) if "msa_rows" in g.files else 2000, int(g["msa_seed"])))
    eng = Engine("cuda:0", max(L, 64), 3000)
    eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    P = n + 1
    print(name, "floor", ["%.1e" % x for x in (g["noise_ca_pass"] if "noise_ca_pass" in g.files else [])])
    for mode in (0, 1, 2):
        eng.set_option("conv_mode", mode)
        c, f = eng.predict(alnmat, None, n, m)
        eng.sync_check()
        ca = eng.fetch("ca_pass", P * L * 3).cpu().numpy().reshape(P, L, 3)
        means = eng.fetch("conf_means", P).cpu().numpy()
        print(" mode", mode, "final %.2e dconf %.1e" % (rmsd(c.cpu().numpy()[:, 1], g["coords"][:, 1]),
                                                      np.abs(f.cpu().numpy() - g["confs"]).max()),
              "per pass", ["%.1e" % rmsd(ca[p], g["ca_pass"][p]) for p in range(P)],
              "dmeans %.1e" % np.abs(means - g["conf_mean_pass"]).max())
    eng.close()
