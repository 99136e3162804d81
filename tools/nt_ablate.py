#!/usr/bin/env python3
"""What bounds the K loop of the 256x128 gather-GEMM? The same launch shapes alone (tools/pipe_probe.py's harness, patch kernel off)
with pieces of the loop compiled OUT (tools/nt_ablate_build.sh: CPCSV_PROBE bits 16 = no MFMAs, 32 = no LDS fragment reads,
64 = no LDS-DMA staging behind the prologue; results are garbage, only the time is read). GPU box:
    bash tools/nt_ablate_build.sh   (CPU container)   ;   python tools/nt_ablate.py > profiles/r04_nt_ablate.txt"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "tools", "probe", "_build")
VARIANTS = [(0, "full kernel"), (16, "no MFMA"), (32, "no fragment reads"), (64, "no staging"), (48, "no MFMA, no reads (staging + barriers)"),
            (80, "no MFMA, no staging (reads + barriers)"), (96, "no reads, no staging (MFMA + barriers)"), (112, "barriers only")]
if os.environ.get("NT_ABLATE_FIXED"):      # what the fixed part (prologue + epilogue) of a launch is made of
    VARIANTS = [(0, "full kernel"), (112, "barriers only (no MFMA / reads / staging)"), (113, "... and no output stores"),
                (114, "... and no epilogue at all"), (1, "full, no output stores"), (2, "full, no epilogue"), (4, "no K loop"),
                (6, "no K loop, no epilogue")]
rows, names = {}, []
for v, what in VARIANTS:
    env = dict(os.environ)
    if v:
        env["CPCSV_LIB_PATH"] = os.path.join(BUILD, "libcpcsv_p%d.so" % v)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pipe_probe.py")], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        m = re.match(r"(.{30}) (\(.*?\))\s+mtile\s+(\d+)\s+([\d.]+) us", line)
        if m:
            name = m.group(1).strip() + " " + m.group(2) + " tile %s" % m.group(3)
            if name not in names:
                names.append(name)
            rows.setdefault(name, {})[v] = float(m.group(4))
print("# tools/nt_ablate.py: launch time in us, one shape alone (30 back-to-back launches), streaming kernel (patch off)")
print("%-62s" % "shape" + "".join("%9s" % ("p%d" % v) for v, _ in VARIANTS))
for name in names:
    print("%-62s" % name + "".join("%9.1f" % rows[name].get(v, float("nan")) for v, _ in VARIANTS))
print()
for v, what in VARIANTS:
    print("p%-4d %s" % (v, what))
