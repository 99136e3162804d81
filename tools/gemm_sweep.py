#!/usr/bin/env python3
"""Tile-config x split-K sweep of gemm_nt on dense stand-ins of the layer shapes (one process per config because the
forced config is read once): python tools/gemm_sweep.py  -> table of microseconds."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(960, 992, 7936), (3840, 496, 3968), (15360, 248, 2048), (3840, 496, 15872 // 4), (960, 2048, 9216), (960, 1481, 8928),
          (15360, 128, 3968), (61440, 128, 4096), (15360, 512, 4096)]
CFGS = {0: "128x128", 1: "128x64", 3: "64x128", 4: "256x128", 5: "64x64"}
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
    import torch
    from cpcsv import kernels as K, _lib as L
    for (m, n, k) in SHAPES:
        x = torch.randn(m, k, device="cuda").bfloat16()
        wt = torch.randn(n, k, device="cuda").bfloat16()
        y = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        out = []
        for sp in (1, 2, 4, 8):
            d = K.gemm_desc(x, wt, y, dtype=L.BF16, M=m, N=n, Cs=k, ldb=k, ldc=n, taps=[(0, 0, 0)], MH=1, MW=1, IH=1, IW=1)
            if sp > 1:
                ws = torch.empty(sp, m, n, device="cuda")
                d.splitk, d.ws, d.ldws, d.ws_rows = sp, ws.data_ptr(), n, m
            for _ in range(3):
                K.gemm_nt(d)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                K.gemm_nt(d)
            e.record()
            torch.cuda.synchronize()
            out.append(s.elapsed_time(e) / 20 * 1e3)
        print("%-8s M=%6d N=%5d K=%6d  " % (CFGS[int(os.environ["CPCSV_NT_FORCE"])], m, n, k) + "  ".join("s%d %6.1f" % (sp, t) for sp, t in zip((1, 2, 4, 8), out)), flush=True)
else:
    for c in CFGS:
        env = dict(os.environ, CPCSV_NT_FORCE=str(c))
        subprocess.run([sys.executable, __file__, "child"], env=env, check=False)
