#!/usr/bin/env python
"""GPU box: time the training-side slice (train.hip) - one residual block's convolution backward at L = 300 and at the
reference's 350 crop - and price its kernels against the f32 matrix-core peak.

    python tools/time_bwd.py [L ...]            (rocprofv3 --kernel-trace --stats around it gives the per-kernel table)

FLOPs (dense, algorithmic): forward = dgrad = wgrad = 2 * 512 * 128 * 25 * L^2.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dmpfold2_amd import synth                       # noqa: E402
from abi import Stages                               # noqa: E402

PEAK = 157.3
Ls = [int(a) for a in sys.argv[1:]] or [300, 350]
st = Stages(synth.synth_weights(0, coord_scale=5.0), max_L=max(Ls), max_N=8)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    for _ in range(reps):
        fn()
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for L in Ls:
    rng = np.random.default_rng(L)
    x = st.to(rng.standard_normal((128, L, L)).astype(np.float32))
    du = st.to(rng.standard_normal((128, L, L)).astype(np.float32))
    dout = st.to(rng.standard_normal((128, L, L)).astype(np.float32))
    u, idx = st.conv_winners(3, x)
    flop = 2.0 * 512 * 128 * 25 * L * L
    t_f = timed(lambda: st.conv_winners(3, x))
    t_b = timed(lambda: st.conv_bwd(3, x, du, idx))
    t_b0 = timed(lambda: st.conv_bwd(3, x, du, None))
    t_n = timed(lambda: st.norm_bwd(3, u, dout))
    print(f"L={L}: forward + winners {t_f:.3f} ms ({flop / t_f / 1e9:.1f} TFLOP/s = {flop / t_f / 1e9 / PEAK:.2f} of the f32 MFMA peak); "
          f"backward with saved winners (pad + db + dgrad + wgrad + reduce) {t_b:.3f} ms ({2 * flop / t_b / 1e9:.1f} TFLOP/s = "
          f"{2 * flop / t_b / 1e9 / PEAK:.2f}); backward recomputing the forward {t_b0:.3f} ms ({3 * flop / t_b0 / 1e9:.1f} TFLOP/s = "
          f"{3 * flop / t_b0 / 1e9 / PEAK:.2f}); norm + scSE + residual backward {t_n:.3f} ms "
          f"(reads u, dout three times + writes du: {7 * 128 * L * L * 4 / t_n / 1e9:.2f} TB/s)", flush=True)
st.eng.sync_check()
ws = st.eng.get_option("device_mib")
print("workspace: one allocation at the first call, for max_L =", max(Ls))
