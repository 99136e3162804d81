#!/usr/bin/env python3
"""The 256x128 streaming gather-GEMM alone, per launch shape (patch kernel off), with a checksum of the output: run once with
CPCSV_NT_PIPE=0 and once with =1 and compare times and checksums (the two main loops must agree bit for bit).
    CPCSV_NT_PIPE=1 python tools/pipe_probe.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch  # noqa: E402
import patch_probe as P  # noqa: E402
from cpcsv import kernels as K  # noqa: E402

if __name__ == "__main__":
    print("# CPCSV_NT_PIPE=%s" % os.environ.get("CPCSV_NT_PIPE", "0"))
    for kind, n, hw, cin, cout, name in P.CASES + [("conv4", 120, 32, 128, 256, "wide conv4 64->32"), ("conv4", 120, 16, 256, 512, "conv4 32->16"),
                                                   ("sub", 120, 8, 1024, 512, "up2 fwd"), ("sub", 120, 4, 2048, 1024, "up1 fwd")]:
        torch.manual_seed(1)
        d, flops, keep = P.case(kind, n, hw, cin, cout)
        d.patch = -1
        t = P.timeit(d)
        y = keep[2]
        torch.cuda.synchronize()
        chk = int(y.view(torch.int16).to(torch.int64).sum().item())
        print("%-30s %-28s mtile %3d  %7.1f us (%4.0f TF)  checksum %d" % (name, str((n, hw, cin, cout)), K.gemm_mtile(d), t, flops / t / 1e6, chk))
