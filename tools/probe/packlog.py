import os, sys, types, collections
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "cpcstoryvisualization-pytorch_amd"))
import torch, bench
from cpcsv import graphs, runtime, kernels as K
runtime.set_compute_dtype("bf16")
bench.pororo_cfg(12, 60)
import trainer as T
torch.manual_seed(0)
tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
tr.setup()
stb, imb = bench.synthetic_batches(12, 60, 1, "cuda")
graphs.PAUSED[0] = True
for _ in range(3):
    tr.train_step(stb, imb)
torch.cuda.synchronize()
log = collections.Counter()
names = {}
for key, net in zip(("G", "D_im", "D_st", "D_se"), tr.nets):
    for n, p in net.named_parameters():
        names[p.data_ptr()] = key + "." + n
import traceback
orig = K._call
def spy(fn, *a):
    if "pack" in fn or "adam" in fn or "copy" in fn or "fill" in fn or "scale" in fn or "colsum" in fn:
        fr = [f for f in traceback.extract_stack() if "cpcstoryvisualization-pytorch_amd/" in f.filename][-5:-1]
        log[(fn, " <- ".join("%s:%d" % (f.filename.split("amd/")[-1], f.lineno) for f in reversed(fr)))] += 1
    log[("ALL", fn)] += 1
    return orig(fn, *a)
K._call = spy
tr.train_step(stb, imb)
torch.cuda.synchronize()
for k, n in sorted(log.items(), key=lambda kv: -kv[1]):
    print(n, k)
