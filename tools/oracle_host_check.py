#!/usr/bin/env python3
"""Does the CPU oracle's 3-step run of the steps3 fixture depend on the HOST (thread count)? Per step: the losses st_G / G_loss of the
oracle in fp32 and in fp64 (fp64 restarted from the fp32 state of that step, like the lock-step test), against the values the
REFERENCE recorded in the fixture, for several torch thread counts.   python tools/oracle_host_check.py [threads ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def run(threads):
    from tests import golden_util as gu, parity_util as pu
    from oracle.cpcsv_oracle import NoiseTape, train_step
    torch.set_num_threads(threads)
    fx3 = gu.load("steps3_plain.npz")
    fx = gu.load(str(fx3["meta/weights_from"]))
    oc, st, sds = pu.oracle_state_for(fx, gu.cfg_of(fx3))
    for k in range(3):
        pre = "s%d/" % k
        stb, imb = gu.batches(fx3, pre)
        tape = gu.noise_tape(fx3, pre)
        snap = pu.oracle_snapshot(st)
        ref = train_step(st, stb, imb, noise=NoiseTape(tape))
        r64 = pu.oracle_step_fp64(oc, snap, stb, imb, tape)
        gl2 = lambda a, b: (sum(float(((a[n].double() - b[n].double()) ** 2).sum()) for n in b) / sum(float((b[n].double() ** 2).sum()) for n in b)) ** 0.5
        print("threads %3d step %d: st_G fp32 %.6f fp64 %.6f reference %.6f | G_loss fp32 %.5f fp64 %.5f reference %.5f | D_st grad fp32 vs fp64 %.2e"
              % (threads, k, float(ref["st_G"]), float(r64["st_G"]), float(fx3[pre + "scalar/st_G"]), float(ref["G_loss"]), float(r64["G_loss"]),
                 float(fx3[pre + "scalar/G_loss"]), gl2(ref["grads_D_st"], r64["grads_D_st"])), flush=True)


if __name__ == "__main__":
    for t in ([int(a) for a in sys.argv[1:]] or [8, 32, os.cpu_count() or 1]):
        run(t)
