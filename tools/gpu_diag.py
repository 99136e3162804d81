#!/usr/bin/env python3
"""First-light diagnostic for a GPU box: runs every op case and the step parity in both dtypes,
never stops at the first failure, writes gpurun_out/diag.txt."""
import os
import sys
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
out = open(os.path.join(REPO, "gpurun_out", "diag.txt"), "w")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.write(s + "\n"); out.flush()


import torch  # noqa: E402
from tests import op_cases, parity_util  # noqa: E402

log("device", torch.cuda.get_device_name(0))
only = sys.argv[1:] or None
for dtype in ("fp32", "bf16"):
    for name in sorted(op_cases.CASES):
        if only and name not in only and "ops" not in only:
            continue
        try:
            rep = op_cases.run_case(name, dtype)
            bad = {k: v for k, v in rep.items() if not (v < (op_cases.tolerances(dtype)[1 if k.startswith("d") else 0]))}
            log("OP", dtype, name, "OK" if not bad else "BAD", {k: "%.2e" % v for k, v in rep.items()})
        except Exception:
            log("OP", dtype, name, "EXC", traceback.format_exc().strip().splitlines()[-1], "|", [l.strip() for l in traceback.format_exc().splitlines() if "cpcsv" in l or "tests/" in l][-2:])
if not only or "step" in only:
    for dtype in ("fp32", "bf16"):
        for tag in ("plain", "cascade"):
            try:
                rep = parity_util.run_step_parity(tag, dtype, check=False)
                log("STEP", dtype, tag, {k: "%.2e" % v for k, v in rep.items()})
            except Exception:
                log("STEP", dtype, tag, "EXC", traceback.format_exc().strip().splitlines()[-1], "|", [l.strip() for l in traceback.format_exc().splitlines() if "cpcsv" in l or "tests/" in l or "trainer" in l or "model" in l][-3:])
