#!/usr/bin/env python3
"""Per-kernel sums of whatever counters one rocprofv3 --pmc pass collected.   python tools/pmc_generic.py <results.db> [top=20]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = db.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
                  "group by kernel_name, counter_name").fetchall()
agg, names = {}, []
for k, c, n, v, dur in rows:
    a = agg.setdefault(k, {"n": n, "dur": dur})
    a[c] = v
    if c not in names:
        names.append(c)
print("%-58s %7s %9s " % ("kernel", "calls", "avg_us") + " ".join("%16s" % c[:16] for c in names) + "   (per-launch averages)")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"])[:top]:
    name = re.sub(r"\(anonymous namespace\)::", "", k)
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))[:58]
    print("%-58s %7d %9.1f " % (name, a["n"], a["dur"] / a["n"] / 1e3) + " ".join("%16.4e" % (a.get(c, 0.0) / a["n"]) for c in names))
