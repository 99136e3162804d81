#!/usr/bin/env python3
"""Measured bf16 errors of one step on the tiny-width golden fixtures (what tests/test_gpu_step.py::test_step_bf16_within_band
bounds): losses, whole-gradient relative L2 per net, worst per-element error / tensor max per net; several seeds of the
non-deterministic (atomic) reductions are NOT involved - the run is in deterministic mode like the test."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
from tests import parity_util as pu  # noqa: E402

for tag in ("plain", "cascade", "seq"):
    rep = pu.run_step_parity(tag, "bf16", check=False)
    keys = [k for k in rep if k.startswith(("loss_rel", "acc_abs", "gradl2_", "grad_"))]
    print(tag, {k: float("%.3g" % rep[k]) for k in keys})
    for k in rep:
        if k.startswith("worst_") and not k.startswith("worst_top") and not k.startswith("worst_loss"):
            print("   ", k, rep[k])
