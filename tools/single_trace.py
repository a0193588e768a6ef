#!/usr/bin/env python
"""GPU box: latency of ONE north-star prediction on one engine (the reference's own use case: one CLI
call) and, under `rocprofv3 --kernel-trace --output-format csv`, where its time goes.

    python tools/single_trace.py run [L N iters minsteps reps]        # timed predictions (prints ms)
    python tools/single_trace.py analyse <kernel_trace.csv>           # timeline of the LAST prediction
"""
import csv
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(L=300, N=2000, iters=10, minsteps=100, reps=4):
    import numpy as np
    import torch
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Engine, encode_aln
    eng = Engine("cuda:0", L, N)
    eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()})
    msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, 0))).to("cuda:0")
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.predict_device(msa, None, iters, minsteps)
        eng.sync_check()
        print("prediction %d: %.2f ms" % (r, (time.perf_counter() - t0) * 1e3), flush=True)
        time.sleep(0.05)          # a visible gap in the trace between predictions


def analyse(path):
    rows = []
    with open(path) as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    # predictions are separated by >= 30 ms without kernels: take the last burst
    cut = 0
    last_end = rows[0][1]
    for i, (s, e, _) in enumerate(rows):
        if s - last_end > 30e6:
            cut = i
        last_end = max(last_end, e)
    rows = rows[cut:]
    t0, t1 = rows[0][0], max(e for _, e, _ in rows)
    print("last prediction: %d kernels, span %.2f ms" % (len(rows), (t1 - t0) / 1e6))
    by = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        by[n][0] += 1
        by[n][1] += e - s
    # time during which no kernel runs, attributed to the kernel that ends the gap
    gaps = defaultdict(lambda: [0, 0])
    cover_end = rows[0][0]
    busy = 0
    for s, e, n in rows:
        if s > cover_end:
            gaps[n][0] += 1
            gaps[n][1] += s - cover_end
            busy += e - s
        else:
            busy += max(0, e - cover_end)
        cover_end = max(cover_end, e)
    idle = (t1 - t0) - busy
    print("some kernel running %.2f ms, none running %.2f ms" % (busy / 1e6, idle / 1e6))
    print("%-46s %7s %10s %9s | gaps before it: %6s %9s" % ("kernel", "calls", "total ms", "avg us", "count", "total ms"))
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
        g = gaps.get(n, [0, 0])
        print("%-46s %7d %10.3f %9.2f | %21d %9.3f" % (n[:46], c, t / 1e6, t / c / 1e3, g[0], g[1] / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*[int(x) for x in sys.argv[2:]])
    else:
        analyse(sys.argv[2])
