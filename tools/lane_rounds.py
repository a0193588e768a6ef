#!/usr/bin/env python
"""Per engine and target: first and last convolution (ms) from a DMP_BATCH_LANE_TRACE / lane-trace interval file."""
import sys
import numpy as np
iv = np.load(sys.argv[1])
S = int(iv[:, 2].max()) + 1
for k in range(S):
    e = iv[iv[:, 2] == k]
    starts = [0] + [i + 1 for i in range(len(e) - 1) if e[i + 1, 0] - e[i, 1] > 30.0] + [len(e)]
    print("engine", k, " ".join(f"[{e[a, 0]:.0f}..{e[b - 1, 1]:.0f} n={b - a}]" for a, b in zip(starts[:-1], starts[1:])))
allv = iv[np.argsort(iv[:, 0])]
end = allv[0, 1]
gaps = []
for r in allv[1:]:
    if r[0] - end > 2.0:
        gaps.append((round(float(end)), round(float(r[0] - end), 1), int(r[2])))
    end = max(end, r[1])
print("gaps > 2 ms (at ms, length, next engine):", gaps)
