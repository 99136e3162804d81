#!/usr/bin/env python3
"""Loss trajectories of a long run, bf16 mode against fp32 mode (same seeds, same rotating synthetic batches, default launch
mode with captured graphs):   python tools/long_run.py run fp32 1000 out_fp32.json ; ... run bf16 ... ; python tools/long_run.py
cmp out_fp32.json out_bf16.json > profiles/r02_bf16_vs_fp32_1000steps.txt.  A GAN on random data is chaotic: individual steps
decorrelate after a few dozen iterations in ANY two runs that differ in round-off, so the comparison is of windowed means."""
import json
import os
import sys

if sys.argv[1] == "run":
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    if "torch" not in sys.modules:
        os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = "0"
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
    import types
    import torch
    import bench
    from cpcsv import runtime
    dtype, steps, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    runtime.set_compute_dtype(dtype)
    bench.pororo_cfg(12, 60)
    import trainer as T
    torch.manual_seed(0)
    tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
    tr.setup()
    batches = [bench.synthetic_batches(12, 60, 100 + i, "cuda") for i in range(8)]
    torch.manual_seed(1)
    torch.cuda.manual_seed_all(1)
    keys = ["seg_D/loss", "img_D/loss", "st_D/loss", "G/loss", "G/im", "G/st", "G/se", "G/im_KL", "G/st_KL"]
    hist = []
    pend = []
    for i in range(steps):
        stb, imb = batches[i % 8]
        nb = batches[(i + 1) % 8]
        o = tr.train_step(stb, imb, next_batches=nb)
        pend.append(torch.stack([torch.as_tensor(o[k], device="cuda", dtype=torch.float32).reshape(()) for k in keys]))
        if len(pend) == 50 or i == steps - 1:
            hist += torch.stack(pend).cpu().tolist()
            pend = []
    json.dump({"keys": keys, "dtype": dtype, "hist": hist}, open(out, "w"))
    print("done", dtype, steps, "last G/loss", hist[-1][3])
else:
    a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    keys = a["keys"]
    n = min(len(a["hist"]), len(b["hist"]))
    win = 100
    print("# %s vs %s, %d steps, ST=12 IM=60 cfg/final.yml widths, 8 rotating synthetic batches, same seeds; means over windows of %d steps"
          % (a["dtype"], b["dtype"], n, win))
    print("# per key: %s-mean / %s-mean (relative difference)" % (a["dtype"], b["dtype"]))
    import math
    bad = sum(1 for h in (a["hist"][:n] + b["hist"][:n]) for v in h if not math.isfinite(v))
    print("# non-finite values: %d" % bad)
    for w0 in range(0, n, win):
        w1 = min(n, w0 + win)
        cells = []
        for j, k in enumerate(keys):
            ma = sum(h[j] for h in a["hist"][w0:w1]) / (w1 - w0)
            mb = sum(h[j] for h in b["hist"][w0:w1]) / (w1 - w0)
            cells.append("%s %.3f/%.3f (%+.1f%%)" % (k, ma, mb, 100.0 * (mb - ma) / (abs(ma) + 1e-12)))
        print("steps %4d-%4d: " % (w0, w1 - 1) + "  ".join(cells))
    print("first 5 steps, G/loss: " + "  ".join("%.4f/%.4f" % (a["hist"][i][3], b["hist"][i][3]) for i in range(5)))
