#!/usr/bin/env python
"""GPU box: do two stages of different targets really run side by side?

Times `stage A alone`, `the f16x3 convolution alone` and both together on two streams (two
contexts).  If the stages share the CUs the joint time approaches the larger of the two, if one
excludes the other it approaches their sum.

    python tools/corun_probe.py [gru_vertical|spd_inverse|eigh|trunk_norm]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dmpfold2_amd import synth, _lib                 # noqa: E402
from dmpfold2_amd.predict import Engine, encode_aln  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "gru_vertical"
L, N = 300, 2000
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
pa = int(os.environ.get('CORUN_PRIO_A', '0')); pb = int(os.environ.get('CORUN_PRIO_B', '0'))
sa, sb = torch.cuda.Stream(dev, priority=pa), torch.cuda.Stream(dev, priority=pb)
ea, eb = Engine(dev, L, N, stream=sa), Engine(dev, L, N, stream=sb)
ea.set_weights(sd)
eb.set_weights(sd)
if os.environ.get("CORUN_CONV_MODE"):
    eb.set_option("conv_mode", int(os.environ["CORUN_CONV_MODE"]))
lib = ea.lib
msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, seed=3))).to(dev)
out = torch.empty(L, 512, device=dev)
cov = torch.eye(21 * L, device=dev) * 2 + 0.001 * torch.rand(21 * L, 21 * L, device=dev)
cov = (cov + cov.t()).contiguous()
work = cov.clone()
M = torch.randn(L, L, device=dev)
M = (M + M.t()).contiguous()
mds = torch.empty(L, 8, device=dev)
x = torch.randn(128, L, L, device=dev)
u = torch.empty(128, L, L, device=dev)
st = torch.empty(128, 2, dtype=torch.float64, device=dev)


def stage_a(reps):
    for _ in range(reps):
        if what == "gru_vertical":
            _lib.check(lib.dmp_gru_vertical(ea.ctx, msa.data_ptr(), N, L, out.data_ptr(), ea.stream()))
        elif what == "spd_inverse":
            with torch.cuda.stream(sa):
                work.copy_(cov)
            _lib.check(lib.dmp_spd_inverse(ea.ctx, work.data_ptr(), 21 * L, ea.stream()))
        elif what == "eigh":
            _lib.check(lib.dmp_eigh_top8(ea.ctx, M.data_ptr(), L, mds.data_ptr(), ea.stream()))
        else:
            raise SystemExit("unknown stage")


def convs(n):
    ms = C.c_float()
    _lib.check(lib.dmp_time_conv5x5(eb.ctx, 1, L, n, C.byref(ms), eb.stream()))
    return ms.value


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


_lib.check(lib.dmp_block_conv5x5_maxout(eb.ctx, 1, x.data_ptr(), L, u.data_ptr(), st.data_ptr(), eb.stream()))
stage_a(1)
torch.cuda.synchronize()
reps = {"gru_vertical": 4, "spd_inverse": 8, "eigh": 40}[what]
ta = timed(lambda: stage_a(reps))
nconv = max(10, int(ta / (2.3 if os.environ.get('CORUN_CONV_MODE') == '1' else 0.73)))
# dmp_time_conv5x5 synchronises its stream itself, so the joint run issues stage A first (asynchronous)
tb = timed(lambda: convs(nconv))
# joint run: dmp_time_conv5x5 returns when its stream is done, so the host clock at its return is the
# end of the convolutions; the synchronize in `timed` gives the end of stage A
conv_end = []


def joint():
    stage_a(reps)
    t0 = time.perf_counter()
    convs(nconv)
    conv_end.append((time.perf_counter() - t0) * 1e3)


tj = timed(joint)
print(f"{what}: alone {ta:.1f} ms, {nconv} convolutions alone {tb:.1f} ms, together {tj:.1f} ms "
      f"(sum {ta + tb:.1f}, max {max(ta, tb):.1f}); in the joint run the convolutions took {conv_end[0]:.1f} ms")
