#!/usr/bin/env python3
"""The shader clock while the gather-GEMM runs: per block, shader cycles (s_memtime) against the 100 MHz real-time counter
(s_memrealtime), for the full kernel and for its ablations (tools/nt_ablate_build.sh with VARIANTS="8 104 56 88 120": stamps +
{MFMA only, staging only, reads only, barriers only}). If the loop's parts add up in TIME because the clock drops when they run
together, it shows here.   python tools/nt_clock.py > profiles/r04_nt_clock.txt   (GPU box)"""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "tools", "probe", "_build")
VARIANTS = [(8, "full kernel"), (104, "MFMA + barriers"), (56, "staging + barriers"), (88, "fragment reads + barriers"), (120, "barriers only")]
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(REPO, "tools"))
    sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
    import torch
    import patch_probe as P
    from cpcsv import _lib as L, kernels as K
    lib = L.load()
    lib.cpcsv_probe_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 8)()
    for kind, n, hw, cin, cout, name in [("sub", 120, 32, 256, 128, "up4 fwd"), ("sub", 120, 16, 512, 256, "up3 fwd"), ("subd", 120, 16, 256, 512, "up3 dgrad"),
                                         ("sub", 120, 4, 2048, 1024, "up1 fwd")]:
        d, flops, keep = P.case(kind, n, hw, cin, cout)
        d.patch = -1
        for _ in range(3):
            K.gemm_nt(d)
        torch.cuda.synchronize()
        lib.cpcsv_probe_read(buf, 1)
        t = P.timeit(d, reps=20)
        lib.cpcsv_probe_read(buf, 0)
        cyc, blocks, real = int(buf[3]), int(buf[4]), int(buf[6])
        print("%-12s %8.1f us  blocks %6d  cycles/block %9.0f  us/block %7.2f  clock %5.2f GHz" % (name, t, blocks // 23, cyc / max(blocks, 1),
              real / max(blocks, 1) / 100.0, cyc / max(real, 1) * 0.1))
    sys.exit(0)
for v, what in VARIANTS:
    env = dict(os.environ, CPCSV_LIB_PATH=os.path.join(BUILD, "libcpcsv_p%d.so" % v))
    print("## %s (CPCSV_PROBE=%d)" % (what, v))
    sys.stdout.flush()
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
