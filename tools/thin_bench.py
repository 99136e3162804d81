#!/usr/bin/env python3
"""Per-kernel timing of the streaming (thin) layers at the benchmark's shapes, thin kernels vs the general gather-GEMM:
algorithmic HBM bytes / time against the 8 TB/s peak (MI355X_MICROARCH.md; 6.29 TB/s measured copy).
   python tools/thin_bench.py > profiles/r02_thin_layers.txt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from cpcsv import functional as F, kernels as K, modules as M, runtime  # noqa: E402

runtime.set_compute_dtype("bf16")
dev = "cuda"
N = int(os.environ.get("THIN_N", "60"))


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2] * 1e3      # us, median


def layer(cin, cout, k, s, p, act, sn=False):
    mods = [M.Conv2d(cin, cout, k, s, p, bias=False, spectral=sn), act]
    net = M.FusedSequential(*mods).to(dev)
    for q in net.parameters():
        q.data.normal_(0, 0.02)
    return net


print("# N=%d frames, bf16, median of 30 launches (torch.cuda.Event); bytes = algorithmic (each tensor touched once)" % N)
print("%-28s %-6s %10s %10s %10s %8s" % ("layer/pass", "path", "us", "MB", "TB/s", "of 8TB/s"))
cases = [("img 128->3 3x3 @64 +tanh", 128, 3, 3, 1, 1, nn.Tanh(), 64),
         ("img_seg 64->1 3x3 @64 +tanh", 64, 1, 3, 1, 1, nn.Tanh(), 64),
         ("D.enc0 3->124 k4s2 @64 +lrelu", 3, 124, 4, 2, 1, nn.LeakyReLU(0.2), 64)]
for name, cin, cout, k, s, p, act, hw in cases:
    for thin in (True, False):
        F._THIN = thin
        net = layer(cin, cout, k, s, p, act)
        cs = (cin + 7) // 8 * 8
        x = torch.randn(N, hw, hw, cs, device=dev).bfloat16()
        if cs != cin:
            x[..., cin:] = 0
        x.requires_grad_(True)
        lay = net._plan()[0]
        w = lay.holder.master()
        with torch.no_grad():
            fwd_us = timed(lambda: net(x))
        y = net(x)
        dy = torch.randn_like(y)
        oh = hw // s
        cso = (cout + 7) // 8 * 8
        in_mb, out_mb = N * hw * hw * cs * 2 / 1e6, N * oh * oh * cso * 2 / 1e6

        def bwd():
            w.grad = None
            x.grad = None
            y.backward(dy, retain_graph=True)
        bwd_us = timed(bwd, iters=15)
        tag = "thin" if thin else "gemm"
        print("%-28s %-6s %10.1f %10.1f %10.2f %8.3f" % (name + " fwd", tag, fwd_us, in_mb + out_mb, (in_mb + out_mb) / fwd_us, (in_mb + out_mb) / fwd_us / 8))
        # backward = act_bwd + dgrad + wgrad + unpack: bytes = dy, y (act), dz r/w, x read (wgrad), dx write
        bmb = 3 * out_mb + out_mb + in_mb + in_mb
        print("%-28s %-6s %10.1f %10.1f %10.2f %8.3f" % (name + " bwd(all)", tag, bwd_us, bmb, bmb / bwd_us, bmb / bwd_us / 8))
F._THIN = True
