#!/usr/bin/env python3
"""Which aten ops does one train step launch (fills, copies, cats ...)? torch.profiler over one step, grouped by op + shapes."""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
import bench
from cpcsv import runtime
runtime.set_compute_dtype("bf16")
bench.pororo_cfg(12, 60)
import trainer as T
torch.manual_seed(0)
tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
tr.setup()
stb, imb = bench.synthetic_batches(12, 60, 1, "cuda")
for _ in range(3):
    tr.train_step(stb, imb)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.train_step(stb, imb)
torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name.split("::")[1] in ("zero_", "fill_", "zeros", "zeros_like", "copy_", "clone", "cat", "add", "add_", "mul", "contiguous", "ones_like", "sum", "mean", "to", "_to_copy", "div", "gt", "randn", "normal_", "empty_strided"):
        st = [f for f in (ev.stack or []) if "cpcstory" in f or "trainer" in f or "miscc" in f]
        key = (ev.name, str(ev.input_shapes)[:60], st[0][-70:] if st else "(autograd/engine)")
        rows[key] = rows.get(key, 0) + 1
for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:70]:
    print("%4d  %-18s %-60s %s" % (v, k[0], k[1], k[2]))
