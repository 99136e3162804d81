#!/usr/bin/env python3
"""Loss histories of the tiny fixture with each capture-once piece switched on alone (debugging aid)."""
import os as _os

# ROCm 7.2 hipGraph "packet capture" corrupts earlier graphs once a process holds ~2900 kernel nodes (see
# cpcsv/graphs.many_graphs_safe); the switch is read when the HIP runtime initialises, i.e. before torch touches the GPU
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import conftest  # noqa
import torch
from tests import golden_util as gu, parity_util as pu

def run(flags, steps=6):
    for k in ("CPCSV_NOGRAD_GRAPH", "CPCSV_CRITIC_GRAPH", "CPCSV_G_GRAPH", "CPCSV_SCORE_GRAPH"):
        os.environ[k] = "1" if k in flags else "0"
    os.environ["CPCSV_GRAPH"] = "0"
    fx = gu.load("step_plain.npz"); oc = gu.cfg_of(fx)
    sds = {k: gu.group(fx, "before/" + k) for k in ("G", "D_im", "D_st", "D_se")}
    tr = pu.make_trainer(oc, sds, "fp32")
    stb, imb = pu.to_dev(gu.batches(fx)[0]), pu.to_dev(gu.batches(fx)[1])
    torch.manual_seed(321); torch.cuda.manual_seed_all(321)
    h = []
    for _ in range(steps):
        out = tr.train_step(stb, imb)
        cs = lambda t: float(t.double().abs().sum())
        h.append(tuple([float(out[k]) for k in ("G/loss", "G/im", "G/st")] +
                       [cs(tr._buckets[k].flat) for k in ("im", "st", "se", "G")] +
                       [cs(torch.cat([p.detach().flatten() for p in n.parameters()])) for n in tr.nets[1:3]]))
    torch.cuda.synchronize()
    return h

os.environ["CPCSV_DBG_G"] = ""
ALL = ("CPCSV_NOGRAD_GRAPH", "CPCSV_CRITIC_GRAPH", "CPCSV_G_GRAPH", "CPCSV_SCORE_GRAPH")
for flags in ((), ALL):
    h = run(flags, 8)
    print("+".join(f.replace("CPCSV_", "").replace("_GRAPH", "") for f in flags) or "eager")
    for t in h[3:]:
        print("      " + " ".join("%9.5f" % v for v in t[:7]))
