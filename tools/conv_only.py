#!/usr/bin/env python
"""Developer helper for PMC passes: run only the pair-trunk convolutions (one trunk pass = 16 launches of the 5x5
kernel of the chosen arithmetic, plus the small norm kernels between them) a few times, plus kernels with exactly known
HBM traffic to calibrate FETCH_SIZE / WRITE_SIZE on this chip:
  * a 256 MiB torch fill (write only, 16 B per lane),
  * a 256 MiB torch copy (read + write, 16 B per lane).
    rocprofv3 --pmc ... --kernel-trace --output-format csv -d out -- python tools/conv_only.py [passes] [L] [conv_mode]
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

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bands = int(sys.argv[4]) if len(sys.argv) > 4 else 0           # option "conv_tile_bands": 0 by length, 1 = 16 x 16, 2 = 8 x 16
st = Stages(synth.synth_weights(0, coord_scale=5.0), L, 8)
st.eng.set_option("conv_mode", mode)
st.eng.set_option("conv_tile_bands", bands)
dev = st.dev

# calibration kernels
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)       # 256 MiB
big.fill_(1.0)
big2 = torch.empty_like(big)
big2.copy_(big)
torch.cuda.synchronize()
z0 = torch.randn(384, L, L, device=dev)
dmap = torch.full((L, L), -1.0, device=dev)
st.trunk_pass(z0, dmap)                              # warm
ms = st.conv_ms(z0, dmap, passes)
print(f"conv5x5 L={L} conv_mode={mode} bands={bands}: {ms:.3f} ms per launch, {2.0 * 128 * 512 * 25 * L * L / ms / 1e9:.1f} TFLOP/s")
