#!/usr/bin/env python3
"""Read traffic (2 x FETCH_SIZE KiB, gfx950 correction) per launch GEOMETRY from a rocprofv3 --pmc FETCH_SIZE rocpd sqlite: one line
per (kernel, grid) - which launch shapes of the GEMM family pull how many bytes through the fabric.
   python tools/pmc_by_grid.py <fetch results.db> [name-substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
if gx is None:
    print("columns of counters_collection:", cols)
    sys.exit(1)
gy, gz = gx.replace("x", "y"), gx.replace("x", "z")
wx = "workgroup_size_x" if "workgroup_size_x" in cols else ("workgroup_x" if "workgroup_x" in cols else "1")
q = ("select kernel_name, %s, %s, %s, %s, count(*), avg(value), avg(duration), sum(value) from counters_collection "
     "where counter_name='FETCH_SIZE' and kernel_name like ? group by kernel_name, %s, %s, %s order by sum(value) desc" % (gx, gy, gz, wx, gx, gy, gz))
print("%-52s %10s %7s %12s %9s %12s" % ("kernel", "blocks", "calls", "read_MB/call", "avg_us", "read_MB_total"))
for name, x, y, z, w, n, v, dur, tot in db.execute(q, ("%" + pat + "%",)):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))[:52]
    blocks = (x // max(w, 1)) * y * z
    print("%-52s %10d %7d %12.2f %9.1f %12.1f" % (name, blocks, n, 2.0 * v * 1024 / 1e6, dur / 1e3, 2.0 * tot * 1024 / 1e6))
