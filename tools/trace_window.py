#!/usr/bin/env python
"""GPU box: print a compact timeline of a rocprofv3 kernel trace around the N-th launch of a kernel.

    python tools/trace_window.py DIR vgru_step_kernel 30000 4.0
prints every kernel that overlaps [t - 1 ms, t + span ms] (t = start of that launch), runs of the
same kernel on the same queue collapsed into one line.
"""
import csv
import glob
import os
import sys


def main():
    d, pat, nth, span = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        short = r["Kernel_Name"].split("(")[0].replace("dmp::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
    rows.sort()
    hits = [r for r in rows if pat in r[2]]
    t = hits[min(nth, len(hits) - 1)][0]
    lo, hi = t - 1_000_000, t + int(span * 1e6)
    sel = [r for r in rows if r[1] >= lo and r[0] <= hi]
    out = []
    for a, b, n, q in sel:
        if out and out[-1][2] == n and out[-1][3] == q and a - out[-1][1] < 200_000:
            out[-1][1] = b
            out[-1][4] += 1
            out[-1][5] += b - a
        else:
            out.append([a, b, n, q, 1, b - a])
    for a, b, n, q, cnt, busy in out:
        print(f"{(a - t) / 1e3:10.1f} .. {(b - t) / 1e3:10.1f} us  q{q:>3s}  x{cnt:<5d} busy {busy / 1e3:9.1f} us  {n}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
