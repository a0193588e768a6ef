#!/usr/bin/env python
"""GPU box: configs[1] prefix (L=200, N=1000, 2 iterations, no minimiser) - HIP path against the CPU oracle, and the
oracle against itself at other thread counts (the case is expansive: which deviations are the oracle's own?)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dmpfold_oracle as O                       # noqa: E402
from dmpfold2_amd import synth                   # noqa: E402
from abi import Stages                           # noqa: E402
from conftest import ca_rmsd                     # noqa: E402

sd = synth.synth_weights(0, coord_scale=5.0)
ow = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
msa = O.encode_aln(synth.synth_msa(200, 1000, 11))
st = Stages(sd, max_L=200, max_N=1000)
outs = {}
for mode in (0, 1):
    st.eng.set_option("conv_mode", mode)
    c, f = st.eng.predict(msa, None, 2, 0)
    st.eng.sync_check()
    outs[mode] = (c.cpu().numpy(), f.cpu().numpy())
refs = {}
for nt in (torch.get_num_threads(), 3, 5):
    torch.set_num_threads(nt)
    rc, rf = O.predict(msa, ow, None, 2, 0, "canonical")
    refs[nt] = (np.asarray(rc), np.asarray(rf))
base = list(refs)[0]
for nt in list(refs)[1:]:
    print("oracle %d vs %d threads: CA-RMSD %.2e  max|dconf| %.2e" % (base, nt, ca_rmsd(refs[base][0][:, 1], refs[nt][0][:, 1]), np.abs(refs[base][1] - refs[nt][1]).max()))
for mode in (0, 1):
    for nt in refs:
        print("HIP conv_mode %d vs oracle %d threads: CA-RMSD %.2e  max|dconf| %.2e" % (mode, nt, ca_rmsd(outs[mode][0][:, 1], refs[nt][0][:, 1]), np.abs(outs[mode][1] - refs[nt][1]).max()))
