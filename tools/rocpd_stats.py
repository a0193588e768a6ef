#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, min, max, share) from a rocprofv3 rocpd SQLite
database (`rocprofv3 --kernel-trace ...` without `--output-format csv` writes *_results.db).

    python tools/rocpd_stats.py gpurun_out/prof/r01_results.db > profiles/<name>.txt
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), "
                   "max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}")
print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print("%-72s %8s %12s %12s %10s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-72s %8d %12.3f %12.2f %10.2f %10.2f %7.2f" % (r[0][:72], r[1], r[2] / 1e6, r[3] / 1e3,
                                                            r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
