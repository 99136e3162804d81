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
                for rpt in range(int(os.environ.get("DIAG_REPEAT", "1"))):
                    rep = parity_util.run_step_parity(tag, dtype, check=False, return_names=True)
                    log("STEP", dtype, tag, {k: ("%.2e" % v if not isinstance(v, str) else v) for k, v in rep.items()})
            except Exception:
                log("STEP", dtype, tag, "EXC", traceback.format_exc().strip().splitlines()[-1], "|", [l.strip() for l in traceback.format_exc().splitlines() if "cpcsv" in l or "tests/" in l or "trainer" in l or "model" in l][-3:])

if only and "bf16full" in only:
    # bf16 vs fp32 of the PRODUCT at full cfg/final.yml widths, same weights/batch/noise, one step
    import types
    sys.path.insert(0, REPO)
    import bench
    from cpcsv import runtime
    res = {}
    for dtype in ("fp32", "bf16"):
        runtime.set_compute_dtype(dtype)
        bench.pororo_cfg(12, 60)
        import trainer as T
        torch.manual_seed(0)
        tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
        tr.setup()
        stb, imb = bench.synthetic_batches(12, 60, 1, "cuda")
        torch.manual_seed(7)
        traj = []
        for it in range(int(os.environ.get("DIAG_STEPS", "3"))):
            o = tr.train_step(stb, imb)
            traj.append({k: float(v) for k, v in o.items() if "Acc" not in k})
        res[dtype] = traj
        del tr
        torch.cuda.empty_cache()
    for it in range(len(res["fp32"])):
        a, b = res["fp32"][it], res["bf16"][it]
        log("FULL step", it, {k: "%.4f/%.4f(%.1e)" % (a[k], b[k], abs(a[k] - b[k]) / (abs(a[k]) + 1e-8)) for k in a})

fmt = lambda rep: {k: ("%.2e" % v if isinstance(v, (int, float)) else v) for k, v in rep.items()}
if only and "fullwidth" in only:
    # one step at cfg/final.yml widths (ST=2/IM=10) vs the ORACLE, both dtypes (tests/test_gpu_fullsize.py)
    from tests.test_gpu_fullsize import fullwidth_vs_oracle
    for dtype in ("fp32", "bf16"):
        try:
            log("FULLWIDTH", dtype, fmt(fullwidth_vs_oracle(dtype)))
        except Exception:
            log("FULLWIDTH", dtype, "EXC", traceback.format_exc())
if only and "multistep" in only:
    for tag, dtype, lock in (("plain", "fp32", True), ("cascade", "fp32", True), ("plain", "fp32", False), ("plain", "bf16", True)):
        try:
            for k, rep in enumerate(parity_util.run_multistep_parity(tag, dtype, lockstep=lock, check=False)):
                log("MULTISTEP", tag, dtype, "lockstep" if lock else "free", k, fmt(rep))
        except Exception:
            log("MULTISTEP", tag, dtype, lock, "EXC", traceback.format_exc())
for tagx in ("clevr", "seq"):
    if only and tagx in only:
        for dtype in ("fp32", "bf16"):
            try:
                log("STEP", dtype, tagx, fmt(parity_util.run_step_parity(tagx, dtype, check=False, return_names=True)))
            except Exception:
                log("STEP", dtype, tagx, "EXC", traceback.format_exc())
