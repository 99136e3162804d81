"""Does the HOST block anywhere in a data-parallel step? One train step of the world-1 RCCL rehearsal (CPCSV_FORCE_EXCHANGE=1, bench
widths) with wall-clock timers around the calls that could wait for the device: every gradient exchange, collective, optimiser step,
graph piece. Round 5: no call takes more than 0.5 ms and the step's host time is 5.3 ms - the lost overlap of the generator's forward
with the critics in exchange mode is NOT the host waiting.   python tools/host_block_probe.py   (GPU box)"""
import os
import sys
import time
import types

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ["CPCSV_FORCE_EXCHANGE"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch, bench
from cpcsv import runtime, dist as cdist
runtime.set_compute_dtype("bf16")
bench.pororo_cfg(12, 60)
import trainer as T
import torch.distributed as dist
torch.manual_seed(0)
tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
tr.setup()
stb, imb = bench.synthetic_batches(12, 60, 1, "cuda")
log = []
def wrap(obj, name, label=None):
    orig = getattr(obj, name)
    def f(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); log.append((label or name, (time.perf_counter() - t0) * 1e3)); return r
    setattr(obj, name, f)
wrap(tr, "_exchange_and_step"); wrap(tr, "_critic_backward"); wrap(tr, "_generator_forward"); wrap(tr, "_nograd_fakes"); wrap(tr, "_prepack_critic")
wrap(dist, "all_reduce", "dist.all_reduce")
import cpcsv.dist as CD
wrap(CD.GradBucket, "reduce_extra_async"); wrap(CD.GradBucket, "allreduce_mean")
import cpcsv.optim as O
wrap(O.FusedAdam, "step", "opt.step"); wrap(O.FusedAdam, "flush_stashes")
for _ in range(8):
    tr.train_step(stb, imb)
torch.cuda.synchronize(); log.clear()
t0 = time.perf_counter()
tr.train_step(stb, imb)
host = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
print("host time of one train_step: %.2f ms" % host)
for n, ms in log:
    print("  %-24s %7.3f ms" % (n, ms))
cdist.shutdown()
