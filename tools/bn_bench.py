#!/usr/bin/env python3
"""The BatchNorm streaming kernels ALONE on the step's activation shapes (bf16, NHWC rows x channels): forward apply, backward
reduce, backward apply - microseconds and achieved HBM TB/s of their compulsory traffic.   python tools/bn_bench.py   (GPU box)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch  # noqa: E402
from cpcsv import _lib as L, kernels as K  # noqa: E402

COPIES = L.BN_SUM_COPIES


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


SHAPES = [("G up4 64x64x128 (120 img, 2 passes)", 491520, 128, 2), ("G up4_seg 64x64x64", 491520, 64, 2), ("G up3 32x32x256", 122880, 256, 2),
          ("G up2 16x16x512", 30720, 512, 2), ("G up1 8x8x1024", 7680, 1024, 2), ("D enc1 16x16x248 (120 img)", 30720, 248, 2),
          ("D enc2 8x8x496", 7680, 496, 2), ("D enc3 4x4x992", 1920, 992, 2), ("G fc 120 rows x 16384 (BatchNorm1d)", 120, 16384, 2)]
print("%-40s %10s %22s %22s %22s" % ("shape", "MB/tensor", "fwd apply us (TB/s)", "bwd reduce us (TB/s)", "bwd apply us (TB/s)"))
SHAPES += [("text BN1d 120 rows x 372 fp32 (tanh)", 120, 372, 2, torch.float32, L.ACT_TANH), ("text BN1d 120 rows x 1095 fp32 (tanh)", 120, 1095, 2, torch.float32, L.ACT_TANH),
           ("text BN1d 24 rows x 248 fp32", 24, 248, 2, torch.float32, L.ACT_RELU)]
for shape in SHAPES:
    name, rows, c, ng = shape[:4]
    tdt = shape[4] if len(shape) > 4 else torch.bfloat16
    actc = shape[5] if len(shape) > 5 else L.ACT_RELU
    cs = (c + 7) // 8 * 8
    x = torch.randn(rows, cs, device="cuda").to(tdt)
    dy = torch.randn(rows, cs, device="cuda").to(tdt)
    y, dz = torch.empty_like(x), torch.empty_like(x)
    pstride = (4 + 2 * COPIES) * cs
    bnbuf = torch.zeros(ng, 4 + 2 * COPIES, cs, device="cuda")
    bnbuf[:, 1] = 1.0
    bnbuf[:, 2] = 1.0
    gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    dgb = torch.zeros(2, c, device="cuda")
    cum = [0, rows // 2, rows] if ng == 2 else [0, rows]
    g = K.bn_groups(cum, pstride)
    mb = rows * cs * x.element_size() / 1e6
    t_f = timeit(lambda: K.bn_apply(x, y, bnbuf[0, 2], bnbuf[0, 3], rows, c, cs, actc, groups=g))
    t_r = timeit(lambda: K.bn_bwd_reduce(dy, x, bnbuf[0, 0], bnbuf[0, 1], gamma, beta, bnbuf[0, 4:], rows, c, cs, actc, groups=g))
    t_a = timeit(lambda: K.bn_bwd_apply(dy, x, dz, bnbuf[0, 0], bnbuf[0, 1], gamma, beta, bnbuf[0, 4:], dgb[0], dgb[1], rows, c, cs, actc,
                                        accumulate=1, groups=g))
    print("%-40s %10.1f %14.1f (%5.2f) %14.1f (%5.2f) %14.1f (%5.2f)" % (name, mb, t_f, 2 * mb / t_f, t_r, 2 * mb / t_r, t_a, 3 * mb / t_a))
