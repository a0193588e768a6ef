#!/usr/bin/env python
"""GPU box: the blocked Gauss-Jordan inverse (dmp_spd_inverse; reference predict.py:53, torch.inverse of the regularised
covariance) at D = 21 L for L = 300, 500, 1000: time of its three forms - block steps in pairs (option gj_pairs, the default), single steps with the look-ahead
(gj_lookahead), single steps in order, the rate
against the f32 MFMA peak, bit-identity of the forms, residual |A inv(A) - I| against a float64 solve.

    python tools/time_inverse.py [L ...]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmpfold2_amd import _lib                            # noqa: E402
from dmpfold2_amd.predict import Engine                  # noqa: E402

Ls = [int(a) for a in sys.argv[1:]] or [300, 500, 1000]
dev = torch.device("cuda:0")
for L in Ls:
    D = 21 * L
    eng = Engine(dev, L, 8, stream=torch.cuda.Stream(dev))
    g = torch.Generator(device="cpu").manual_seed(L)
    X = torch.randn(D, 2 * D if D <= 8000 else D // 2, generator=g)
    A = (X @ X.T / X.shape[1] + 4.5 * torch.eye(D) / np.sqrt(8.0)).float().to(dev)     # covariance-like + the reference's ridge
    del X
    res = {}
    # (pairs, look-ahead, blocked diagonal sweep); "blocked" = option gj_diag_blocked = 1 with the pairs (round 5; not the default)
    modes = {"pairs": (1, 0, 0), "lookahead": (0, 2, 0), "serial": (0, 0, 0), "blocked": (1, 0, 1)}
    for name, (pairs, la, blocked) in modes.items():
        eng.set_option("gj_diag_blocked", blocked)
        eng.set_option("gj_pairs", pairs)
        eng.set_option("gj_lookahead", la)                # 2: at every size (1 = the library's size policy)
        times = []
        for rep in range(6):
            a = A.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(eng._stream):
                _lib.check(eng.lib.dmp_spd_inverse(eng.ctx, a.data_ptr(), D, eng.stream()))
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        res[name] = (a, min(times[1:]))
    same = all(bool(torch.equal(res["serial"][0], res[m][0])) for m in ("pairs", "lookahead"))
    inv = res["pairs"][0]
    sub = slice(0, min(D, 2048))
    err = float(((A[sub].double() @ inv.double())[:, sub] - torch.eye(D, device=dev, dtype=torch.float64)[sub, sub]).abs().max())
    sym = float((inv - inv.T).abs().max())
    flop = float(D) ** 3                                  # symmetric Gauss-Jordan: half of the 2 D^3 of the full one
    dch = float((res["blocked"][0] - inv).abs().max()) / float(inv.abs().max())
    for name in modes:
        ms = res[name][1]
        print(f"L={L} D={D} {name:9s}: {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s (lower triangle, D^3) = "
              f"{flop / ms / 1e9 / 157.3:.3f} of the f32 MFMA peak", flush=True)
    print(f"L={L} D={D}: pairs == look-ahead == serial bitwise: {same}; |A inv - I| max {err:.2e} (first 2048 rows); "
          f"|inv - inv^T| max {sym:.1e}; blocked sweep against the chain of 128 pivots: max |d| / max |inv| = {dch:.1e}", flush=True)
    eng.close()
    del A, res, inv
    torch.cuda.empty_cache()
