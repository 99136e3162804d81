#!/usr/bin/env python3
"""Micro-benchmark of gemm_nt on a few shapes with diagnostic flags (what bounds an iteration?)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
from cpcsv import kernels as K, _lib as L

dev = "cuda"
def run(name, n, h, w, cin, cout, k=3, splitk=1, reps=20):
    x = torch.randn(n, h, w, cin, device=dev).bfloat16()
    wt = torch.randn(cout, k * k * cin, device=dev).bfloat16()
    y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16)
    taps = [(u - k // 2, v - k // 2, u * k + v) for u in range(k) for v in range(k)]
    m = n * h * w
    for dbg in (0,):
        d = K.gemm_desc(x, wt, y, dtype=L.BF16, M=m, N=cout, Cs=cin, ldb=wt.shape[1], ldc=cout, taps=taps, MH=h, MW=w, IH=h, IW=w)
        ws = None
        if splitk > 1:
            ws = torch.empty(splitk, m, cout, device=dev)
            d.splitk, d.ws, d.ldws, d.ws_rows = splitk, ws.data_ptr(), cout, m
        for _ in range(3): K.gemm_nt(d)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): K.gemm_nt(d)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / reps * 1e3
        fl = 2.0 * m * cout * k * k * cin
        tiles = ((m + 127) // 128) * ((cout + 127) // 128)
        nk = k * k * ((cin + 63) // 64)
        print("%-28s dbg=%d  %8.1f us  %7.1f TF/s  tiles=%d x split %d, %d K-tiles/block -> %.2f us per K-tile per block-wave"
              % (name, dbg, us, fl / us / 1e6, tiles, splitk, nk // splitk, us / max(1, (tiles * splitk + 511) // 512) / (nk // splitk)))

run("up3-like 61440x256 K4608", 60, 32, 32, 512, 256)
run("up2-like 15360x512 K9216", 60, 16, 16, 1024, 512)
run("head 960x992 K13392", 60, 4, 4, 1488, 992)
run("head 960x992 K13392 split8", 60, 4, 4, 1488, 992, splitk=8)
run("enc3-like 960x992 K7936 s8", 60, 4, 4, 496, 992, k=4, splitk=8) if False else None
