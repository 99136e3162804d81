#!/usr/bin/env python3
"""How well-conditioned is ONE training step of the steps3 fixtures? The oracle against ITSELF: from the state before each of the
three steps, the step is repeated with every weight multiplied by (1 + eps * N(0,1)) - eps = 1e-7 is one fp32 ulp - and each net's
whole gradient is compared with the unperturbed run (relative L2). A smooth step answers eps-sized (1e-6); a step with a
pre-activation within eps of a LeakyReLU / ReLU kink behind a 3-sample BatchNorm answers with a jump. CPU only.
usage: python tools/oracle_conditioning.py [plain|cascade] [eps] [trials]"""
import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from tests import golden_util as gu
    from tests import parity_util as pu
    from oracle.cpcsv_oracle import NoiseTape, make_state, train_step
    tag = sys.argv[1] if len(sys.argv) > 1 else "plain"
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-7
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    fx3 = gu.load("steps3_%s.npz" % tag)
    fx = gu.load(str(fx3["meta/weights_from"]))
    oc, st, _ = pu.oracle_state_for(fx, gu.cfg_of(fx3))

    def restore(snap):
        s2 = make_state(oc)
        for (nsd, osd), net, opt in zip(snap, (s2.netG, s2.netD_im, s2.netD_st, s2.netD_se), (s2.optG, s2.optD_im, s2.optD_st, s2.optD_se)):
            net.load_state_dict(copy.deepcopy(nsd))
            opt.load_state_dict(copy.deepcopy(osd))
        return s2

    print("# %s fixture, %d threads, weights * (1 + %.0e * N(0,1)): relative L2 of each net's gradient against the unperturbed oracle step"
          % (tag, torch.get_num_threads(), eps))
    jumps = 0
    for k in range(int(fx3["meta/steps"])):
        pre = "s%d/" % k
        stb, imb = gu.batches(fx3, pre)
        tape = gu.noise_tape(fx3, pre)
        snap = pu.oracle_snapshot(st)
        ref = train_step(st, stb, imb, noise=NoiseTape(tape))
        for trial in range(trials):
            g = torch.Generator().manual_seed(trial)
            s2 = restore(snap)
            with torch.no_grad():
                for net in (s2.netG, s2.netD_im, s2.netD_st, s2.netD_se):
                    for p in net.parameters():
                        p.mul_(1 + eps * torch.randn(p.shape, generator=g))
            r2 = train_step(s2, stb, imb, noise=NoiseTape(tape))
            out = []
            for key, gk in pu.NETKEYS:
                num = sum(float(((r2[gk][n].double() - v.double()) ** 2).sum()) for n, v in ref[gk].items())
                den = sum(float((v.double() ** 2).sum()) for n, v in ref[gk].items())
                e = (num / den) ** 0.5
                jumps += e > 1e-4
                out.append("%s %.2e" % (key, e))
            print("step %d trial %d: " % (k, trial) + "  ".join(out), flush=True)
    print("# (net, step, trial) triples that answered with a jump (> 1e-4): %d of %d" % (jumps, 4 * trials * int(fx3["meta/steps"])))


if __name__ == "__main__":
    main()
