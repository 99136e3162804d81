#!/usr/bin/env python3
"""Per-kernel matrix-core utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE): MFMA-busy share = MFMA busy cycles / (GRBM_GUI_ACTIVE x 1024 SIMDs) when the counter
is summed over SIMDs; printed raw as well so the ratio can be re-derived (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES
counts cycles, SQ_WAVE_CYCLES quad-cycles).   python tools/pmc_mfma.py <results.db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
                  "group by kernel_name, counter_name").fetchall()
agg = {}
for k, c, n, v, dur in rows:
    a = agg.setdefault(k, {"n": n, "dur": dur})
    a[c] = v
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE (one pass); sums over launches")
print("%-64s %7s %10s %14s %14s %14s %10s" % ("kernel", "calls", "avg_us", "MFMA_BUSY", "GUI_ACTIVE", "WAVE_CYCLES", "mfma/act"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"])[:28]:
    name = re.sub(r"\(anonymous namespace\)::", "", k)
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))[:64]
    mf, ga, wc = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0), a.get("SQ_WAVE_CYCLES", 0.0)
    # MFMA busy is summed over the chip's 1024 SIMDs (256 CUs x 4); GUI_ACTIVE is one chip-level clock count
    frac = mf / (ga * 1024.0) if ga else float("nan")
    print("%-64s %7d %10.1f %14.3e %14.3e %14.3e %10.3f" % (name, a["n"], a["dur"] / a["n"] / 1e3, mf, ga, wc, frac))
