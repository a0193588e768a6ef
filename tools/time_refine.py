#!/usr/bin/env python
"""GPU box: minimiser timing, cluster (16 workgroups) against the single-workgroup kernel."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dmpfold2_amd import synth      # noqa: E402
from abi import Stages              # noqa: E402

sd = synth.synth_weights(0, coord_scale=5.0)
st = Stages(sd, max_L=1024, max_N=8)
rng = np.random.default_rng(3)
for L, steps in ((82, 100), (200, 100), (300, 100), (500, 200), (1000, 1000)):
    step = rng.standard_normal((L, 3))
    step *= 3.8 / np.linalg.norm(step, axis=1, keepdims=True)
    ca = st.to(np.cumsum(step, axis=0).astype(np.float32))
    out = {}
    for single in (0, 1):
        st.eng.set_option("refine_single", single)
        st.refine(ca, steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = st.refine(ca, steps)
        torch.cuda.synchronize()
        out[single] = ((time.perf_counter() - t0) / 3 * 1e3, r.cpu().numpy())
    st.eng.set_option("refine_single", 0)
    print(f"L={L} steps={steps}: cluster {out[0][0]:.3f} ms ({out[0][0] / steps * 1e3:.1f} us/step), "
          f"single {out[1][0]:.3f} ms ({out[1][0] / steps * 1e3:.1f} us/step), max |diff| {np.abs(out[0][1] - out[1][1]).max():.2e}")
st.eng.sync_check()
st.eng.close()
