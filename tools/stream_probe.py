"""Isolated HBM rate of the streaming kernels of the step (BatchNorm forward / backward, fused layer update) at the
step's real tensor shapes: each kernel re-launched alone, HIP events around 30 launches. Compare with the in-step rates of
profiles/r03_roofline.txt section 2 (there the kernels share the GPU with two or three other streams).
usage: python tools/stream_probe.py   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpcstoryvisualization-pytorch_amd"))
import torch
from cpcsv import kernels as K, _lib as L

dev = "cuda"


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


print("rows      C | kernel           us     MB    TB/s  frac(8 TB/s)")
for rows, c in ((122880, 128), (61440, 128), (30720, 256), (30720, 248), (7680, 512), (7680, 496), (1920, 1024), (1920, 992)):
    cs = (c + 7) // 8 * 8
    x = torch.randn(rows, cs, device=dev).bfloat16()
    dy = torch.randn(rows, cs, device=dev).bfloat16()
    y = torch.empty_like(x)
    f = lambda *s: torch.randn(*s, device=dev)
    scale, shift, mean, invstd, gamma, beta = f(cs), f(cs), f(cs), f(cs).abs() + 0.5, f(c), f(c)
    sums = torch.zeros(2 * L.BN_SUM_COPIES, cs, device=dev)
    dgamma, dbeta = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    mb = rows * cs * 2 / 1e6
    for name, fn, tensors in (
            ("bn_apply", lambda: K.bn_apply(x, y, scale, shift, rows, c, cs, L.ACT_RELU), 2),
            ("bn_bwd_reduce", lambda: K.bn_bwd_reduce(dy, x, mean, invstd, gamma, beta, sums, rows, c, cs, L.ACT_RELU), 2),
            ("bn_bwd_apply", lambda: K.bn_bwd_apply(dy, x, y, mean, invstd, gamma, beta, sums, dgamma, dbeta, rows, c, cs, L.ACT_RELU), 3)):
        us = timed(fn)
        print(f"{rows:6d} {c:5d} | {name:14s} {us:7.1f} {tensors * mb:6.1f} {tensors * mb / us:7.2f}  {tensors * mb / us / 8:5.3f}", flush=True)
