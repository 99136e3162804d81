#!/usr/bin/env python3
"""The patch-resident conv main loop against the streaming gather-GEMM, one launch shape at a time, ALONE on the GPU
(30 back-to-back launches, HIP events): the layer shapes of the bench workload that are patch-eligible.
    python tools/patch_probe.py > profiles/r04_patch_probe.txt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from cpcsv import functional as F, kernels as K, _lib as L  # noqa: E402


def tconv_taps(hw):
    geom = F.ConvGeom(4, 2, 1)
    taps, phases = [], []
    for tp, mh, mw, _, sc in geom.dgrad_launches(2 * hw, 2 * hw):
        phases.append((len(taps), len(tp), sc[4], sc[5]))
        taps += tp
    return taps, phases


def case(kind, n, hw, cin, cout):
    cs, cout_s = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
    if kind in ("sub", "tconv"):                    # stride 1 over an hw x hw grid, scattered 2hw x 2hw output
        x = torch.randn(n, hw, hw, cs).to(torch.bfloat16).cuda()
        w = (torch.randn(cout, 16 * cs) * 0.05).to(torch.bfloat16).cuda()
        y = torch.empty(n, 2 * hw, 2 * hw, cout_s, dtype=torch.bfloat16, device="cuda")
        taps, phases = (F.SUB_FWD_TAPS, F.SUB_PHASES) if kind == "sub" else tconv_taps(hw)
        d = K.gemm_desc(x, w, y, dtype=L.BF16, M=n * hw * hw, N=cout, Cs=cs, ldb=16 * cs, ldc=cout_s, taps=taps, MH=hw, MW=hw, IH=hw, IW=hw,
                        scatter=(2 * hw, 2 * hw, 2, 2, 0, 0), phases=phases, act=L.ACT_RELU)
        flops = 2.0 * n * hw * hw * cout * 16 * cs
    else:                                           # 4x4 stride-2 window over a 2hw x 2hw input ("conv4": critic forward; "subd": sub-pixel dgrad)
        x = torch.randn(n, 2 * hw, 2 * hw, cs).to(torch.bfloat16).cuda()
        w = (torch.randn(cout, 16 * cs) * 0.05).to(torch.bfloat16).cuda()
        y = torch.empty(n, hw, hw, cout_s, dtype=torch.bfloat16, device="cuda")
        taps = sorted(F.ConvGeom(4, 2, 1).fwd_taps(), key=lambda t: ((t[0] + 1) & 1, (t[1] + 1) & 1)) if kind == "conv4" else F.SUB_DGRAD_TAPS
        d = K.gemm_desc(x, w, y, dtype=L.BF16, M=n * hw * hw, N=cout, Cs=cs, ldb=16 * cs, ldc=cout_s, taps=taps, MH=hw, MW=hw, IH=2 * hw, IW=2 * hw,
                        sy=2, sx=2)
        flops = 2.0 * n * hw * hw * cout * 16 * cs
    return d, flops, (x, w, y)


def timeit(d, reps=30):
    for _ in range(3):
        K.gemm_nt(d)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        K.gemm_nt(d)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


CASES = [("sub", 120, 32, 256, 128, "up4 fwd"), ("sub", 120, 16, 512, 256, "up3 fwd"), ("sub", 120, 32, 128, 64, "up4_seg fwd"),
         ("sub", 120, 16, 256, 128, "up3_seg fwd"), ("tconv", 120, 16, 248, 124, "critic enc1 dgrad (D step)"),
         ("tconv", 60, 16, 248, 124, "critic enc1 dgrad (G step)"),
         ("subd", 120, 32, 128, 256, "up4 dgrad"), ("subd", 120, 16, 256, 512, "up3 dgrad"), ("subd", 120, 32, 64, 128, "up4_seg dgrad"),
         ("subd", 120, 16, 128, 256, "up3_seg dgrad"), ("conv4", 120, 16, 124, 248, "critic enc1 fwd (D step)"),
         ("conv4", 60, 16, 124, 248, "critic enc1 fwd (G step)")]

if __name__ == "__main__":
    print("# tools/patch_probe.py: one launch shape alone, 30 back-to-back launches; TFLOP/s of the EXECUTED product (16 taps)")
    print("%-30s %-28s %10s %10s %8s" % ("layer", "(n, grid, Cin, Cout)", "stream us", "patch us", "speedup"))
    for kind, n, hw, cin, cout, name in CASES:
        d, flops, keep = case(kind, n, hw, cin, cout)
        d.patch = -1
        t0 = timeit(d)
        d.patch = 1
        t1 = timeit(d)
        print("%-30s %-28s %7.1f (%4.0f TF) %7.1f (%4.0f TF) %6.2fx" % (name, str((n, hw, cin, cout)), t0, flops / t0 / 1e6, t1, flops / t1 / 1e6, t0 / t1))
