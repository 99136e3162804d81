#!/usr/bin/env python3
"""GPU-side duration of the phases of one eager train step (events on the main stream at phase boundaries).
Monkeypatches nothing: re-implements the step's phase boundaries by wrapping trainer methods."""
import os as _os

# ROCm 7.2 hipGraph "packet capture" corrupts earlier graphs once a process holds ~2900 kernel nodes (see
# cpcsv/graphs.many_graphs_safe); the switch is read when the HIP runtime initialises, i.e. before torch touches the GPU
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

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
import miscc.utils as U

torch.manual_seed(0)
tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
tr.setup()
stb, imb = bench.synthetic_batches(12, 60, 1, "cuda")
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


orig_ng = tr._nograd_fakes
def ng(*a):
    mark("start")
    r = orig_ng(*a)
    mark("nograd_done")
    return r
tr._nograd_fakes = ng
orig_gf = tr._generator_forward
def gf(*a):
    mark("gfwd_start")
    r = orig_gf(*a)
    mark("gfwd_done")
    return r
tr._generator_forward = gf
orig_kl = T.KL_loss
def kl(*a):
    if not any(n == "score_done" for n, _ in marks[-3:]):
        mark("score_done")
    return orig_kl(*a)
T.KL_loss = kl
side_marks = []
orig_cb = tr._critic_backward
def cb(key, net, a, tag, feat):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    r = orig_cb(key, net, a, tag, feat)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    side_marks.append((key, e0, e1))
    return r
tr._critic_backward = cb
if os.environ.get("DBG_EXTRA_STREAM") == "1":
    # experiment: ONE tiny launch per critic update on a fifth stream that waits for that critic's stream (what a collective's
    # internal stream does) - does a fifth busy stream alone cost the overlap of the generator's forward with the critics?
    _x5 = torch.cuda.Stream()
    _buf = torch.zeros(64, device="cuda")
    _cb2 = tr._critic_backward
    def cb5(key, net, a, tag, feat):
        r = _cb2(key, net, a, tag, feat)
        side = tr._side_stream(key)
        _x5.wait_stream(side)
        with torch.cuda.stream(_x5):
            _buf.add_(1.0)
        side.wait_stream(_x5)
        return r
    tr._critic_backward = cb5
real_marks = []
orig_cr = tr._critic_real
def cr(key, net, imgs):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    r = orig_cr(key, net, imgs)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    real_marks.append((key, e0, e1))
    return r
tr._critic_real = cr
orig_cs = tr._critic_score
score_marks = []
def cs(key, net, a):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    r = orig_cs(key, net, a)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    score_marks.append((key, e0, e1))
    return r
tr._critic_score = cs
orig_step = tr.optimizerG.step
def gstep(*a, **k):                      # (pending / gscale in the data-parallel form)
    mark("gbwd_done")
    orig_step(*a, **k)
    mark("adam_done")
tr.optimizerG.step = gstep

for _ in range(8):
    tr.train_step(stb, imb)
marks.clear(); side_marks.clear(); score_marks.clear(); real_marks.clear()
N = 10
for _ in range(N):
    tr.train_step(stb, imb)
torch.cuda.synchronize()
import collections
acc = collections.OrderedDict()
names = [n for n, _ in marks]
per = len(marks) // N
for s in range(N):
    seg = marks[s * per:(s + 1) * per]
    for (n0, e0), (n1, e1) in zip(seg, seg[1:]):
        acc[n0 + " -> " + n1] = acc.get(n0 + " -> " + n1, 0.0) + e0.elapsed_time(e1)
    if s + 1 < N:
        acc["adam_done -> next start"] = acc.get("adam_done -> next start", 0.0) + seg[-1][1].elapsed_time(marks[(s + 1) * per][1])
tot = 0
for k, v in acc.items():
    print("%-34s %7.3f ms" % (k, v / N)); tot += v / N
print("sum %.3f ms" % tot)
# side streams relative to the main stream's nograd_done mark of the same step
starts = [e for n, e in marks if n == "nograd_done"]
gd = [e for n, e in marks if n == "gfwd_done"] or starts
for name, lst, per_step in (("critic real pass", real_marks, 3), ("critic fwd+bwd", side_marks, 3), ("critic scoring", score_marks, 3)):
    if not lst:
        continue
    for k in range(per_step):
        a = sum(starts[s].elapsed_time(lst[s * per_step + k][1]) for s in range(N)) / N
        b = sum(starts[s].elapsed_time(lst[s * per_step + k][2]) for s in range(N)) / N
        print("%-16s %-3s starts %+7.3f ms, ends %+7.3f ms after nograd_done" % (name, lst[k][0], a, b))
if len(gd) == N and gd is not starts:
    print("G forward ends   %+7.3f ms after nograd_done" % (sum(starts[s].elapsed_time(gd[s]) for s in range(N)) / N))

