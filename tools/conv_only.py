#!/usr/bin/env python
"""Developer helper for PMC passes: run only the conv5x5 kernel (block 1) a few times, plus three
kernels with exactly known HBM traffic to calibrate FETCH_SIZE / WRITE_SIZE on this chip:
  * a 256 MiB torch fill (write only, 16 B per lane),
  * a 256 MiB torch copy (read + write, 16 B per lane),
  * act_pad_kernel (4 B per lane: reads 128*L*L*4 B, writes 128*P*P*4 B).
    rocprofv3 --pmc ... --kernel-trace --output-format csv -d out -- python tools/conv_only.py [iters] [L]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmpfold2_amd import synth, _lib                 # noqa: E402
from dmpfold2_amd.predict import Engine              # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = Engine("cuda:0", L, 8)
sd = synth.synth_weights(0, coord_scale=5.0)
eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
dev = eng.device

# calibration kernels
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)       # 256 MiB
big.fill_(1.0)
big2 = torch.empty_like(big)
big2.copy_(big)
torch.cuda.synchronize()
x = torch.randn(128, L, L, device=dev)
u = torch.empty(128, L, L, device=dev)
st = torch.empty(128, 2, dtype=torch.float64, device=dev)
_lib.check(eng.lib.dmp_block_conv5x5_maxout(eng.ctx, 1, x.data_ptr(), L, u.data_ptr(), st.data_ptr(),
                                            eng.stream()))
torch.cuda.synchronize()
ms = C.c_float()
_lib.check(eng.lib.dmp_time_conv5x5(eng.ctx, 1, L, iters, C.byref(ms), eng.stream()))
print(f"conv5x5 L={L}: {ms.value:.3f} ms per launch, {2.0 * 128 * 512 * 25 * L * L / ms.value / 1e9:.1f} TFLOP/s")
