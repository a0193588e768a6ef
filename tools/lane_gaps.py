#!/usr/bin/env python
"""GPU box: analyse a rocprofv3 kernel trace (CSV) of bench.py - how much of the wall time has a
conv5x5 kernel running, and which kernels run while none does.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o bench -- python bench.py ...
    python tools/lane_gaps.py DIR
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", d)
        return 1
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            short = name.split("(")[0].replace("dmp::", "").replace("void ", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
    rows.sort()
    # restrict to the timed region: after the last long idle period before the final burst is hard to
    # know; use the second half of the trace (steady state)
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) // 2
    rows = [r for r in rows if r[0] >= t_lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    conv = sorted((a, b) for a, b, n, q in rows if n.startswith("conv5x5"))
    # union of conv intervals
    busy = 0
    gaps = []
    cur_a, cur_b = conv[0]
    for a, b in conv[1:]:
        if a > cur_b:
            busy += cur_b - cur_a
            gaps.append((cur_b, a))
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    wall = t1 - t0
    print(f"window {wall / 1e6:.1f} ms, conv busy {busy / 1e6:.1f} ms = {100.0 * busy / wall:.1f}%  "
          f"({len(conv)} launches, mean {sum(b - a for a, b in conv) / len(conv) / 1e3:.1f} us)")
    hist = defaultdict(float)
    for a, b in gaps:
        g = (b - a) / 1e3
        key = "<20us" if g < 20 else "<100us" if g < 100 else "<1ms" if g < 1000 else "<5ms" if g < 5000 else ">=5ms"
        hist[key] += g / 1e3
    print("lane-idle time by gap length (ms):", {k: round(v, 1) for k, v in hist.items()})
    # which kernels overlap the gaps (time-weighted)
    other = [(a, b, n) for a, b, n, q in rows if not n.startswith("conv5x5")]
    occ = defaultdict(float)
    gi = 0
    gaps.sort()
    import bisect
    starts = [g[0] for g in gaps]
    for a, b, n in other:
        i = max(0, bisect.bisect_left(starts, a) - 1)
        while i < len(gaps) and gaps[i][0] < b:
            lo, hi = max(a, gaps[i][0]), min(b, gaps[i][1])
            if hi > lo:
                occ[n] += hi - lo
            i += 1
    tot_gap = sum(b - a for a, b in gaps)
    print(f"total lane-idle {tot_gap / 1e6:.1f} ms; kernel time inside the idle periods (ms):")
    for n, v in sorted(occ.items(), key=lambda kv: -kv[1])[:14]:
        print(f"   {n:40s} {v / 1e6:8.1f}")
    # per-kernel totals in the window
    tot = defaultdict(lambda: [0, 0.0])
    for a, b, n, q in rows:
        tot[n][0] += 1
        tot[n][1] += b - a
    print("kernel totals in the window (ms):")
    for n, (cnt, v) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {n:40s} {cnt:7d} {v / 1e6:8.1f}  avg {v / cnt / 1e3:8.1f} us")
    return 0


if __name__ == "__main__":
    sys.exit(main())
