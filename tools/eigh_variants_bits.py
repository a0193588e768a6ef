#!/usr/bin/env python
"""GPU box: dmp_eigh_top8 with the three forms of the tridiagonalisation (cluster launch, one launch per Householder
step, single workgroup) on matrix families - are the bits the same?  (round 4: one benchmark target's minimised trace
told the cluster and the per-step form apart.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi                                               # noqa: E402
from dmpfold2_amd import synth                           # noqa: E402

st = abi.Stages(synth.synth_weights(0), max_L=320, max_N=8)
eng = st.eng
rng = np.random.default_rng(0)


def gram(P):
    D = np.linalg.norm(P[:, None] - P[None], axis=2)
    return (0.5 * (D[0:1, :] ** 2 + D[:, 0:1] ** 2 - D ** 2)).astype(np.float32)


fams = {}
for L in (96, 300):
    fams[f"random symmetric L={L}"] = [(lambda a: ((a + a.T) / 2).astype(np.float32))(rng.standard_normal((L, L))) for _ in range(4)]
    fams[f"protein-like Gram L={L}"] = [gram(np.cumsum(rng.standard_normal((L, 3)) * 2.2, axis=0)) for _ in range(4)]
    fams[f"collapsed trace Gram L={L}"] = [gram(rng.standard_normal((L, 3)) * s) for s in (1e-3, 1e-1, 30.0, 1e3)]
    fams[f"mixed-scale Gram L={L}"] = [gram(np.concatenate([rng.standard_normal((L // 2, 3)) * 1e-2, rng.standard_normal((L - L // 2, 3)) * 50])) for _ in range(4)]
    fams[f"rank-1 + noise L={L}"] = [(np.outer(v, v) + 1e-6 * ((lambda a: (a + a.T) / 2)(rng.standard_normal((L, L))))).astype(np.float32)
                                     for v in [rng.standard_normal(L) * 10 for _ in range(4)]]
forms = {"cluster": dict(tridiag_cluster=1, tridiag_single=0), "per step": dict(tridiag_cluster=0, tridiag_single=0),
         "single workgroup": dict(tridiag_cluster=0, tridiag_single=1)}
for name, mats in fams.items():
    eq_cs = eq_c1 = 0
    worst = 0.0
    for M in mats:
        out = {}
        for f, o in forms.items():
            for k, v in o.items():
                eng.set_option(k, v)
            out[f] = st.eigh_top8(st.to(M)).cpu().numpy()
        eng.sync_check()
        eq_cs += int(np.array_equal(out["cluster"], out["per step"]))
        eq_c1 += int(np.array_equal(out["cluster"], out["single workgroup"]))
        worst = max(worst, float(np.abs(out["cluster"] - out["per step"]).max() / max(1e-30, np.abs(out["cluster"]).max())))
    print(f"{name:32s}: cluster == per step on {eq_cs}/{len(mats)} matrices (worst relative difference {worst:.1e}); "
          f"cluster == single workgroup on {eq_c1}/{len(mats)}", flush=True)
