#!/usr/bin/env python3
"""Full-size (ST=12/IM=60) loss histories: eager vs eager (run-to-run spread of the fp32 atomics), eager vs the
captured pieces, fp32 vs bf16 with fixed noise."""
import os, sys, types
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
from tests import test_gpu_fullsize as T, parity_util as pu

KEYS = ("G/loss", "img_D/loss", "st_D/loss", "seg_D/loss", "seg_D/fake", "G/im", "G/st", "G/se")

def run(dtype, graphs, fixed_noise, steps=5):
    for k in ("CPCSV_NOGRAD_GRAPH", "CPCSV_CRITIC_GRAPH", "CPCSV_G_GRAPH", "CPCSV_SCORE_GRAPH"):
        os.environ[k] = "1" if graphs else "0"
    tr, (stb, imb) = T._trainer(dtype)
    if fixed_noise:
        pu.set_noise(tr.nets[0], T._fixed_noise())
    torch.manual_seed(7); torch.cuda.manual_seed_all(7)
    h = []
    for _ in range(steps):
        out = tr.train_step(stb, imb)
        h.append([float(out[k]) for k in KEYS])
    del tr; torch.cuda.empty_cache()
    return h

def run_flags(flags, steps=5):
    for k in ("CPCSV_NOGRAD_GRAPH", "CPCSV_CRITIC_GRAPH", "CPCSV_G_GRAPH", "CPCSV_SCORE_GRAPH"):
        os.environ[k] = "1" if k in flags else "0"
    tr, (stb, imb) = T._trainer("bf16")
    torch.manual_seed(7); torch.cuda.manual_seed_all(7)
    h = []
    for _ in range(steps):
        out = tr.train_step(stb, imb)
        h.append([float(out[k]) for k in KEYS])
    del tr; torch.cuda.empty_cache()
    return h

for flags in ((), ("CPCSV_NOGRAD_GRAPH",), ("CPCSV_CRITIC_GRAPH",), ("CPCSV_G_GRAPH",), ("CPCSV_SCORE_GRAPH",)):
    os.environ["CPCSV_G_WGRAD_BRANCH"] = "1"
    h = run_flags(flags)
    print("+".join(flags) or "eager")
    for row in h[2:]:
        print("    " + " ".join("%9.5f" % v for v in row))
