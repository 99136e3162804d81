#!/usr/bin/env python3
"""Where does the HOST spend its time per step? cProfile over a few bench steps (GPU box)."""
import os as _os

# ROCm 7.2 hipGraph "packet capture" corrupts earlier graphs once a process holds ~2900 kernel nodes (see
# cpcsv/graphs.many_graphs_safe); the switch is read when the HIP runtime initialises, i.e. before torch touches the GPU
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import cProfile
import os
import pstats
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from cpcsv import runtime  # noqa: E402

runtime.set_compute_dtype("bf16")
bench.pororo_cfg(12, 60)
import trainer as T  # noqa: E402
torch.manual_seed(0)
tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
tr.setup()
stb, imb = bench.synthetic_batches(12, 60, 1, "cuda")
import time  # noqa: E402
for _ in range(8):
    tr.train_step(stb, imb)
torch.cuda.synchronize()
# host time inside hipGraphLaunch, per replay
_replay = torch.cuda.CUDAGraph.replay
acc = {"n": 0, "t": 0.0}


def timed_replay(self):
    a = time.perf_counter()
    _replay(self)
    acc["t"] += time.perf_counter() - a
    acc["n"] += 1


torch.cuda.CUDAGraph.replay = timed_replay
N = 10
t0 = time.perf_counter()
for _ in range(N):
    tr.train_step(stb, imb)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step %.2f ms ; wall per step %.2f ms ; graph replays per step %.1f taking %.2f ms of host time"
      % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, acc["n"] / N, acc["t"] / N * 1e3))
torch.cuda.CUDAGraph.replay = _replay
if os.environ.get("HOST_PROFILE_SHORT"):
    sys.exit(0)
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):      # backward on this thread, so cProfile sees it
    tr.train_step(stb, imb)
    pr.enable()
    for _ in range(3):
        tr.train_step(stb, imb)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(40)
