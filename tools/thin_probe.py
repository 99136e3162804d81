#!/usr/bin/env python3
"""Raw-launch timing of thin3x3_fwd at the img shape (60x64x64x128 -> 3): back-to-back launches through the C ABI."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
from cpcsv import kernels as K
N, H, W, CS, CO = 60, 64, 64, int(os.environ.get("CS", "128")), 3
x = torch.randn(N, H, W, CS, device="cuda").bfloat16()
w = torch.randn(CO, 9 * CS, device="cuda").bfloat16()
y = torch.empty(N, H, W, 8, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    K.thin3x3_fwd(x, w, y, N, H, W, CS, CO, 3)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    K.thin3x3_fwd(x, w, y, N, H, W, CS, CO, 3)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 50
mb = (x.numel() * 2 + y.numel() * 2) / 1e6
print("probe=%s R=%s TW=%s CS=%d: %.1f us  %.2f TB/s" % (os.environ.get("CPCSV_THIN_PROBE", "0"), os.environ.get("CPCSV_THIN_R", "auto"), os.environ.get("CPCSV_THIN_TW", "auto"), CS, us, mb / us))
