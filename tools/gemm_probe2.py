#!/usr/bin/env python3
"""Dense (1-tap) gemm_nt timing sweep: time vs split-K and K, to separate per-K-tile cost from fixed cost."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
from cpcsv import kernels as K, _lib as L

dev = "cuda"


def run(m, n, k, splitk=1, reps=30):
    x = torch.randn(m, k, device=dev).bfloat16()
    wt = torch.randn(n, k, device=dev).bfloat16()
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    d = K.gemm_desc(x, wt, y, dtype=L.BF16, M=m, N=n, Cs=k, ldb=k, ldc=n, taps=[(0, 0, 0)], MH=1, MW=1, IH=1, IW=1)
    if splitk > 1:
        ws = torch.empty(splitk, m, n, device=dev)
        d.splitk, d.ws, d.ldws, d.ws_rows = splitk, ws.data_ptr(), n, m
    for _ in range(3):
        K.gemm_nt(d)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        K.gemm_nt(d)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    nk = (k + 63) // 64
    print("M=%6d N=%5d K=%6d split=%2d blocks=%5d ktiles/blk=%4d  %8.1f us  %7.1f TF/s" %
          (m, n, k, splitk, tiles * splitk, nk // splitk, us, 2.0 * m * n * k / us / 1e6))


for sp in (1, 2, 4, 8, 16):
    run(960, 992, 7936, sp)
for sp in (1, 2, 4, 8):
    run(3840, 496, 3968, sp)
for k in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    run(8192, 2048, k)          # 1024 tiles: steady state per-K-tile cost
for k in (1024, 4096):
    run(32768, 1024, k)
