#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace rocpd sqlite as a text Gantt: every kernel of the LAST complete step in start
order with its offset from the step start (us), duration, queue and grid, so that one can read off which chain the step
waits on.   python tools/step_trace.py <results.db> [min_us=0] [step_from_end=1]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start_c = "start" if "start" in cols else "start_timestamp"
end_c = "end" if "end" in cols else "end_timestamp"
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, %s, %s, queue_id" % (start_c, end_c) + ((", %s, %s" % (gx, wx)) if gx and wx else ", 0, 1")
rows = list(db.execute("select %s from kernels order by %s" % (sel, start_c)))
clean = lambda n: re.sub(r"^void ", "", re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n)))
adam_ends = sorted(r[2] for r in rows if "adam_kernel" in r[0])
bounds = adam_ends[3::4]
t0, t1 = bounds[-1 - back], bounds[-back]
qs = sorted({r[3] for r in rows})
print("# step of %.3f ms; columns: offset_us dur_us queue blocks kernel" % ((t1 - t0) / 1e6))
for n, s, e, q, g, w in rows:
    if s < t0 or e > t1 + 1 or (e - s) / 1e3 < min_us:
        continue
    print("%9.1f %7.1f  %s%-2s %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, "   " * qs.index(q), q, (g // max(w, 1)) if g else 0, clean(n)[:70]))
