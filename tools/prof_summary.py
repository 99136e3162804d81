#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite) as a text table:
   python tools/prof_summary.py gpurun_out/prof2/bench_results.db [steps] > profiles/r01_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(top_kernels)")]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from top_kernels")]
tot = sum(r["total_duration"] for r in rows)
print("# rocprofv3 --kernel-trace --stats (rocpd top_kernels view, microseconds) ; %d steps in the trace" % steps)
print("# total kernel time %.3f ms  (%.3f ms per step)" % (tot / 1e3, tot / 1e3 / steps))
print("%-78s %8s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\(.*", "", name)[:78]
    print("%-78s %8d %12.1f %12.2f %6.2f" % (name, r["total_calls"], r["total_duration"] / 1e3, r["average"], r["percentage"]))
