#!/usr/bin/env python3
"""In-kernel time stamps of the rolling thin forward kernel (block 0 and block 100, thread 0; 100 MHz wall clock)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
os.environ["CPCSV_THIN_DBG_PTR"] = str(dbg.data_ptr())
from cpcsv import _lib as L
lib = L.load()
cs = int(os.environ.get("CS", "128")); cout = 3 if cs == 128 else 1
N, H, W = 60, 64, 64
x = torch.randn(N, H, W, cs, device="cuda").bfloat16()
w = (torch.randn(cout, 9 * cs, device="cuda") * 0.05).bfloat16()
y = torch.empty(N, H, W, 8, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    dbg.zero_()
    torch.cuda.synchronize()
    rc = lib.cpcsv_thin3x3_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), N, H, W, cs, cout, 3, st)
    torch.cuda.synchronize()
assert rc == 0
d = dbg.cpu().tolist()
for b in (0, 64):
    t = [v for v in d[b:b + 60] if v]
    print("block", 0 if b == 0 else 100, "stamps (us from kernel start):", " ".join("%.2f" % ((v - t[0]) / 100.0) for v in t))
