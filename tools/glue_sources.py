#!/usr/bin/env python3
"""Which Python lines of the step still launch torch / rocclr kernels (fills, copies, cats, adds ...)? One EAGER train step
(graph pieces paused: the same launches the captures bake in) under torch.profiler with stacks; every device kernel that is not
one of csrc/'s is attributed to the innermost repo frame of the CPU op that launched it. GPU box only."""
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

import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEWS = ("view", "as_strided", "slice", "select", "detach", "expand", "permute", "t.", "transpose", "alias", "empty", "unsqueeze",
         "squeeze", "reshape", "_unsafe_view", "split", "unbind", "narrow", "size", "stride", "is_", "record_stream", "_local_scalar",
         "lift_fresh", "_to_copy_meta", "chunk", "unfold", "set_", "resize", "storage", "_reshape_alias", "new_empty", "sym_",
         "result_type", "can_cast", "equal_meta", "_has_compatible", "is_pinned", "_pin", "contiguous_meta")
sites = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if name.startswith(("cpcsv", "prim")) or any(name.startswith(v) for v in VIEWS):
            return out
        flat = []
        for a in list(args) + list((kwargs or {}).values()) + [out]:
            if torch.is_tensor(a):
                flat.append(a)
            elif isinstance(a, (list, tuple)):
                flat.extend(x for x in a if torch.is_tensor(x))
        if not any(t.is_cuda for t in flat):
            return out
        frames = [f for f in traceback.extract_stack() if "cpcstoryvisualization-pytorch_amd/" in f.filename]
        site = " <- ".join("%s:%d" % (f.filename.split("cpcstoryvisualization-pytorch_amd/")[-1], f.lineno) for f in reversed(frames[-3:]))
        sites[(name, site)] += 1
        return out


with torch.autograd.set_multithreading_enabled(False), Log():
    tr.train_step(stb, imb)
    torch.cuda.synchronize()
print("## aten ops on device tensors in one eager step (views and allocations excluded), by op and call site")
for (name, site), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("%4d  %-22s %s" % (n, name, site))
