#!/usr/bin/env python3
"""One critic's update (forward real+fake, backward, Adam) alone on one stream, repeated: run under
rocprofv3 --kernel-trace and read tools/prof_by_grid.py to see each layer's GEMM launches without the
cross-stream contention of the full step."""
import os
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
from miscc.utils import compute_discriminator_loss  # noqa: E402

torch.manual_seed(0)
tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
tr.setup()
netD = tr.nets[1]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
real = torch.randn(60, 3, 64, 64, device="cuda")
fake = torch.randn(60, 3, 64, 64, device="cuda")
ones, zeros = torch.ones(60, device="cuda"), torch.zeros(60, device="cuda")
labels = (torch.rand(60, 9, device="cuda") > 0.5).float()
from miscc.config import cfg  # noqa: E402
mu = torch.randn(60, cfg.TEXT.DIMENSION + 9 + cfg.GAN.CONDITION_DIM, device="cuda")
import time  # noqa: E402
for i in range(reps + 3):
    if i == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    tr._buckets["im"].zero()
    err = compute_discriminator_loss(netD, real, fake, ones, zeros, labels, mu, 1)[0]
    err.backward()
    tr.im_optimizerD.step()
torch.cuda.synchronize()
print("critic update: %.3f ms per iteration" % ((time.perf_counter() - t0) / reps * 1e3))
