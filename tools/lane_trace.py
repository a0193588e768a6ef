#!/usr/bin/env python
"""GPU box: where does the lane idle?  Runs the throughput pipeline WITHOUT a profiler, with the HIP-event
timing of every convolution launch switched on, and prints how much of the wall time a convolution was
running, the idle time by gap length and the longest gaps.

    python tools/lane_trace.py [streams 4] [targets 16]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dmpfold2_amd import synth, _lib                    # noqa: E402
from dmpfold2_amd.predict import Pipeline, encode_aln    # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L, N = int(os.environ.get("LT_L", "300")), int(os.environ.get("LT_N", "2000"))
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
msas = [torch.from_numpy(encode_aln(synth.synth_msa(L, N, seed=i))).to(dev) for i in range(8)]
pipe = Pipeline(dev, L, N, sd, streams=S)
for e in pipe.engines:                                   # A/B knob
    e.set_option("refine_single", int(os.environ.get("LT_REFINE_SINGLE", "0")))
pipe.run(msas[:S], 10, 100)
torch.cuda.synchronize()
lib = pipe.lib
cap = 176 * (T // S + 4)
for e in pipe.engines:
    _lib.check(lib.dmp_profile_enable(e.ctx, 1, cap))
t0 = time.perf_counter()
pipe.run([msas[i % 8] for i in range(T)], 10, 100)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
iv = []
for k, e in enumerate(pipe.engines):
    a = (C.c_float * cap)()
    b = (C.c_float * cap)()
    n = C.c_int()
    _lib.check(lib.dmp_profile_conv_intervals(e.ctx, pipe.engines[0].ctx, a, b, cap, C.byref(n)))
    iv += [(a[i], b[i], k) for i in range(n.value)]
pipe.close()
iv.sort()
if os.environ.get("LT_SAVE"):
    np.save(os.environ["LT_SAVE"], np.array(iv, dtype=np.float64))
busy = sum(b - a for a, b, _ in iv)
span = iv[-1][1] - iv[0][0]
print(f"S={S}: {T} targets in {wall:.0f} ms = {T / wall * 1e3:.2f} structures/s; {len(iv)} convolutions, "
      f"mean {busy / len(iv):.4f} ms; first to last {span:.0f} ms, lane busy {busy / span:.3f}")
gaps = [(iv[i + 1][0] - max(x[1] for x in iv[max(0, i - 3):i + 1]), iv[i][2], iv[i + 1][2], iv[i][1]) for i in range(len(iv) - 1)]
edges = [0.005, 0.02, 0.05, 0.1, 0.3, 1.0, 3.0, 1e9]
hist = [0.0] * len(edges)
cnt = [0] * len(edges)
for g, *_ in gaps:
    if g <= 0:
        continue
    k = next(i for i, e in enumerate(edges) if g < e)
    hist[k] += g
    cnt[k] += 1
lo = 0.0
for e, h, c in zip(edges, hist, cnt):
    print(f"  gaps {lo:5.3f} .. {e if e < 1e8 else float('inf'):5.3f} ms: {c:5d} gaps, {h:8.1f} ms ({100 * h / span:4.1f} % of the span)")
    lo = e
same = sum(g for g, a, b, _ in gaps if g > 0 and a == b)
print(f"  idle between two convolutions of the SAME engine: {same:.1f} ms; of different engines: "
      f"{sum(g for g, a, b, _ in gaps if g > 0 and a != b):.1f} ms")
print("  longest gaps (ms, engine before -> after, at ms):",
      [(round(g, 2), a, b, round(t)) for g, a, b, t in sorted(gaps, reverse=True)[:12]])
# time by the number of convolutions in flight (sweep over the interval end points)
ev = sorted([(a, 1) for a, b, _ in iv] + [(b, -1) for a, b, _ in iv])
depth, last, at = 0, ev[0][0], {}
for t, d in ev:
    at[depth] = at.get(depth, 0.0) + (t - last)
    depth += d
    last = t
print("  time with k convolutions in flight:", "  ".join(f"{k}: {v:.0f} ms ({100 * v / span:.1f} %)" for k, v in sorted(at.items())))
