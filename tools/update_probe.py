#!/usr/bin/env python3
"""Raw-launch timing of cpcsv_layer_update on one layer shape (default: G.upsample1, sub-pixel 2048->1024)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
from cpcsv import kernels as K, _lib as L, functional as F
cout, cin, taps, S, sub = [int(x) for x in os.environ.get("SHAPE", "1024,2048,9,16,1").split(",")]
dev = "cuda"
G = torch.randn(cout, S * cin, device=dev)
p = torch.randn(cout, cin, taps, device=dev) * 0.02
m = torch.zeros_like(p); v = torch.zeros_like(p)
fwd = torch.empty(cout, S * cin, device=dev, dtype=torch.bfloat16)
bwd = torch.empty(cin, S * cout, device=dev, dtype=torch.bfloat16)
hyper = torch.tensor([1.0, 1e-4], device=dev)
d = L.UpdateDesc()
d.G, d.p, d.m, d.v, d.fwd, d.bwd, d.hyper = G.data_ptr(), p.data_ptr(), m.data_ptr(), v.data_ptr(), fwd.data_ptr(), bwd.data_ptr(), hyper.data_ptr()
d.beta1, d.beta2, d.eps, d.dtype = 0.5, 0.999, 1e-8, L.BF16
d.Cout, d.Cin, d.taps, d.S, d.Cin_s, d.Cout_s, d.sum = cout, cin, taps, S, cin, cout, sub
for i in range(16):
    d.tapmap[i] = i if i < S else -1
    d.masks[i] = F.SUB_MASKS[i] if sub else 0
for _ in range(3):
    K.layer_update(d)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    K.layer_update(d)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 20
n = cout * cin * taps
byt = cout * cin * S * 4 + n * 24 + 2 * cout * cin * S * 2
print("probe=%s shape=%s: %.1f us  %.2f TB/s (all phases' bytes %.0f MB)" % (os.environ.get("CPCSV_UPD_PROBE", "0"), (cout, cin, taps, S), us, byt / us / 1e6, byt / 1e6))
