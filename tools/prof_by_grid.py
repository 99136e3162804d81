#!/usr/bin/env python3
"""Per-launch-geometry durations from a rocprofv3 --kernel-trace rocpd sqlite: one line per (kernel, grid):
   python tools/prof_by_grid.py <results.db> [name-substring] [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
q = ("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration), min(duration), sum(duration) "
     "from kernels where name like ? group by name, grid_x, grid_y, grid_z order by sum(duration) desc")
print("%-52s %10s %7s %9s %9s %10s" % ("kernel", "blocks", "calls/s", "avg_us", "min_us", "ms/step"))
for name, gx, gy, gz, wx, n, avg, mn, tot in db.execute(q, ("%" + pat + "%",)):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))[:52]
    blocks = (gx // max(wx, 1)) * gy * gz
    print("%-52s %10d %7.1f %9.1f %9.1f %10.3f" % (name, blocks, n / steps, avg / 1e3, mn / 1e3, tot / 1e6 / steps))
