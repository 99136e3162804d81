import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "cpcstoryvisualization-pytorch_amd"))
from tests import parity_util as pu
print("# product step vs the REFERENCE's own record (tests/golden/step_<tag>.npz scalar/*, grad/*), no oracle in between; beside it vs the oracle")
for tag in ("plain", "cascade", "clevr", "seq"):
    for dt in ("fp32", "bf16"):
        rep = pu.run_step_parity(tag, dt, check=True)
        v = rep["vs_reference"]
        f = lambda d: " ".join("%s=%.2e" % (k, d[k]) for k in ("loss_rel", "gradl2_G", "gradl2_D_im", "gradl2_D_st", "gradl2_D_se", "acc_abs"))
        print("%-8s %s  vs reference: %s" % (tag, dt, f(v)))
        print("%-8s %s  vs oracle   : %s" % (tag, dt, f(rep)))
