#!/usr/bin/env python3
"""Which CPU ops launch the device kernels of the step that are NOT csrc/'s? tools/glue_sources.py sees the ops Python dispatches;
this one asks torch.profiler, which also sees what the C++ autograd engine launches on its own (gradient sums of a tensor used
twice, zeros for an unused output, the cat behind a split ...). One EAGER train step (graph pieces paused: the captures bake in the
same launches); every foreign kernel is attributed to the chain of CPU ops above its launch. GPU box only."""
import os as _os

_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import collections
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from cpcsv import graphs, runtime  # noqa: E402

runtime.set_compute_dtype("bf16")
bench.pororo_cfg(12, 60)
import trainer as T  # noqa: E402

torch.manual_seed(0)
tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
tr.setup()
stb, imb = bench.synthetic_batches(12, 60, 1, "cuda")
graphs.PAUSED[0] = True
for _ in range(4):
    tr.train_step(stb, imb)
torch.cuda.synchronize()

from torch.profiler import ProfilerActivity, profile  # noqa: E402

with torch.autograd.set_multithreading_enabled(False), profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA],
                                                                with_stack=True) as prof:
    tr.train_step(stb, imb)
    torch.cuda.synchronize()

FOREIGN = ("at::", "rocclr", "Cat", "elementwise", "Functor", "rocprim", "hipcub")
sites = collections.Counter()
for ev in prof.events():
    ks = [k for k in getattr(ev, "kernels", []) if any(f in k.name for f in FOREIGN)]
    if not ks:
        continue
    chain, p = [ev.name], ev.cpu_parent
    while p is not None and len(chain) < 5:
        chain.append(p.name)
        p = p.cpu_parent
    stack = [s for s in (ev.stack or []) if "cpcstoryvisualization-pytorch_amd/" in s][:2]
    stack = [s.split("cpcstoryvisualization-pytorch_amd/")[-1] for s in stack]
    for k in ks:
        sites[(k.name[:60], " <- ".join(chain), " | ".join(stack))] += 1
print("## foreign device kernels of one eager step, by kernel, launching op chain and the nearest repo frames")
for (k, chain, stack), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("%4d  %-60s  %s   [%s]" % (n, k, chain, stack))
print("total", sum(sites.values()))
