#!/usr/bin/env python
"""Per-kernel timeline of the last `steps` block steps of one inverse from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d /tmp/invp -o inv -- python tools/time_inverse.py 300
    python tools/inverse_timeline.py /tmp/invp/*/inv_kernel_trace.csv [first_kernel_index] [count]
"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"].split("(")[0].replace("void dmp::", "").replace("dmp::", "")
        if n.startswith("gj_"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t0 = rows[i0][0]
for s, e, n, q in rows[i0:i0 + cnt]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{q}  {n}")
