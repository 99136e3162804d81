#!/usr/bin/env python3
"""The cpu_baseline leg of bench.py (the oracle = CPU fp32 port of the reference step, bench workload) at several thread counts:
why the bench line uses 32 of the host's cores. One fresh process per count (MKL-DNN primitive caches, thread pools).
    python tools/cpu_threads.py [counts...]      default: 16 32 64 <all cores>   -> profiles/r05_cpu_threads.txt"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = [int(a) for a in sys.argv[1:]] or sorted({16, 32, 64, os.cpu_count() or 1})
print("# oracle train_step, ST=12 / IM=60, cfg/final.yml widths, fp32; host has %d cores; 1 warm-up + 2 timed steps per row" % (os.cpu_count() or 1))
for n in counts:
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench.cpu_baseline(12, 60, threads=%d)))" % (REPO, n))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REPO)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("threads %4d: %7.3f story-frames/s   (%s)" % (n, d["value"], d["sample"].split(": ")[-1]))
    except Exception:
        print("threads %4d: failed: %s" % (n, (out.stderr or out.stdout)[-300:]))
