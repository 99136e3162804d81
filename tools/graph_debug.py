import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
import torch
from tests.test_gpu_graph import _run
he, we, ge, bne = _run(False)
hg, wg, gg, bng = _run(True)
for k in ("seg_D/loss", "img_D/loss", "st_D/loss", "G/loss", "G/im_KL"):
    print(k, ["%.4f/%.4f" % (a[k], b[k]) for a, b in zip(he, hg)])
print("w diff", (we - wg).abs().max().item(), bne[:3], bng[:3])
