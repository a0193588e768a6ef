#!/usr/bin/env python
"""GPU box: what runs inside the front-end hole of the throughput scheduler?

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/lane_trace.py 4 8
    python tools/hole_profile.py DIR

Finds the longest interval without a convolution between the first and the last convolution of the trace and
prints, per kernel and hardware queue, when it first starts and last ends inside that interval (ms after the
interval's start), its launch count and its busy time.
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        short = r["Kernel_Name"].split("(")[0].replace("dmp::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
    rows.sort()
    conv = [r for r in rows if "conv5x5" in r[2]]
    best, end = (0, 0, 0), conv[0][1]
    for r in conv[1:]:
        if r[0] - end > best[0]:
            best = (r[0] - end, end, r[0])
        end = max(end, r[1])
    gap, lo, hi = best
    t00 = conv[0][0]
    # every vertical-GRU chain (launches less than 1 ms apart on one queue) and every interval > 3 ms without a convolution
    chains = []
    for a, b, n, q in rows:
        if "vgru2_step" not in n and "vgru_step" not in n:
            continue
        if chains and chains[-1][3] == q and a - chains[-1][1] < 1_000_000:
            chains[-1][1] = b
            chains[-1][2] += 1
        else:
            chains.append([a, b, 1, q])
    for a, b, cnt, q in chains:
        inside = sum(1 for r in conv if r[1] > a and r[0] < b)
        print(f"chain q{q}: {(a - t00) / 1e6:9.2f} .. {(b - t00) / 1e6:9.2f} ms  {cnt} launches, {(b - a) / 1e3 / cnt:.1f} us per launch, "
              f"{inside} convolutions overlap it")
    end = conv[0][1]
    for r in conv[1:]:
        if r[0] - end > 3_000_000:
            print(f"no convolution: {(end - t00) / 1e6:9.2f} .. {(r[0] - t00) / 1e6:9.2f} ms  ({(r[0] - end) / 1e6:.1f} ms)")
        end = max(end, r[1])
    if os.environ.get("HOLE_BRIEF"):
        return 0
    print(f"longest interval without a convolution: {gap / 1e6:.2f} ms")
    agg = {}
    for a, b, n, q in rows:
        if b < lo or a > hi:
            continue
        k = (n, q)
        e = agg.setdefault(k, [a, b, 0, 0])
        e[0], e[1], e[2], e[3] = min(e[0], a), max(e[1], b), e[2] + 1, e[3] + (min(b, hi) - max(a, lo))
    for (n, q), (a, b, cnt, busy) in sorted(agg.items(), key=lambda kv: kv[1][0]):
        print(f"{(a - lo) / 1e6:8.2f} .. {(b - lo) / 1e6:8.2f} ms  q{q:>3s}  x{cnt:<6d} busy {busy / 1e6:8.2f} ms  {n[:70]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
