#!/usr/bin/env python3
"""bf16 / fp32 product step against the fp64 oracle at cfg/final.yml widths for a chosen batch and model:
   python tools/casc_diag.py <st> <im> <cascade 0|1> [dtypes...]   (the fp64 oracle step is CPU work: minutes at ST=12/IM=60)"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0"); os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = "0"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "cpcstoryvisualization-pytorch_amd"))
import time
import torch
from tests import test_gpu_fullsize as T
st, im, casc = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
t0 = time.time()
kw = {}
if os.environ.get("RECON") is not None:
    kw["reconstruct_loss"] = float(os.environ["RECON"])
T._fullwidth_oracles(st=st, im=im, cascade=casc, **kw)
print("# oracle fp32 + fp64 steps: %.0f s" % (time.time() - t0), flush=True)
for dtype in (sys.argv[4:] or ["fp32", "bf16"]):
    rep = T.fullwidth_vs_oracle(dtype, cascade=casc)
    print("FULLWIDTH st=%d im=%d cascade=%d %s " % (st, im, casc, dtype) + " ".join(
        "%s=%s" % (k, ("%.3g" % v) if isinstance(v, float) else v) for k, v in rep.items() if not k.startswith("worst")), flush=True)
    print("    worst_top_G: " + rep.get("worst_top_G", ""), flush=True)
