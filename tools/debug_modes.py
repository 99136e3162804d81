#!/usr/bin/env python3
"""Generator weight-gradient accumulators of steps 1 and 2, batched halves vs one launch set per half (deterministic, fixed noise)."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "cpcstoryvisualization-pytorch_amd"))
import torch
from tests import test_gpu_fullsize as T
from tests import parity_util as pu
import miscc.utils as MU
from cpcsv import functional as F, runtime
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
runtime.set_deterministic(True)
snaps = {}
for mode in ("batched", "two"):
    MU.BATCH_PASSES = mode == "batched"
    F._PAIR = False
    tr, (stb, imb) = T._trainer(dtype)
    pu.set_noise(tr.nets[0], T._fixed_noise())
    orig = tr.optimizerG.step
    steps = []
    def grab(closure=None, _o=orig, _t=tr):
        names = [l.name for l, w, _ in _t.optimizerG._layers]
        steps.append((names, [l._g.clone() for l, w, _ in _t.optimizerG._layers], _t._buckets["G"].flat.clone()))
        return _o()
    tr.optimizerG.step = grab
    outs = []
    for i in range(2):
        outs.append({k: float(v) for k, v in tr.train_step(stb, imb).items() if "Acc" not in k})
    torch.cuda.synchronize()
    snaps[mode] = (steps, outs)
    del tr
    torch.cuda.empty_cache()
for i in range(2):
    na, a, fa = snaps["batched"][0][i]
    nb, b, fb = snaps["two"][0][i]
    print("step", i, "flat small-param grads rel diff %.3e" % float((fa.double() - fb.double()).norm() / fb.double().norm()))
    for n, x, y in zip(na, a, b):
        print("   %-28s %.3e   |g|=%.3e" % (n, float((x.double() - y.double()).norm() / y.double().norm()), float(y.double().norm())))
    la, lb = snaps["batched"][1][i], snaps["two"][1][i]
    print("   losses:", {k: (round(la[k], 5), round(lb[k], 5)) for k in ("G/loss", "img_D/loss", "st_D/loss", "seg_D/loss")})
