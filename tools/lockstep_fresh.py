#!/usr/bin/env python3
"""Lock-step run of the steps3 fixture (tests/parity_util.run_multistep_parity) with TWO product trainers per step: the one that has
run the earlier steps (and was synchronised from the oracle after each), and a FRESH one built from the oracle's state right before
the step. Equal gradients = no state survives the synchronisation; per tensor of the story critic: relative error against the
oracle and the projection <g_product, g_oracle> / <g_oracle, g_oracle> (a pure scale factor shows as rel = |1 - proj|).
usage: python tools/lockstep_fresh.py [plain|cascade] [oracle threads]"""
import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
if "torch" not in sys.modules:
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
import torch  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "plain"
    if len(sys.argv) > 2:
        torch.set_num_threads(int(sys.argv[2]))
    from cpcsv import runtime
    from tests import golden_util as gu
    from tests import parity_util as pu
    from oracle.cpcsv_oracle import NoiseTape, train_step
    fx3 = gu.load("steps3_%s.npz" % tag)
    fx = gu.load(str(fx3["meta/weights_from"]))
    oc, st, sds = pu.oracle_state_for(fx, gu.cfg_of(fx3))
    runtime.set_deterministic(True)
    tr = pu.make_trainer(oc, sds, "fp32")
    keys = ("G", "D_im", "D_st", "D_se")

    def one(trainer, stb, imb, tape):
        pu.set_noise(trainer.nets[0], pu.TapeSource(tape))
        grads = {}
        hooks = pu._capture_grads(trainer, grads)
        out = trainer.train_step(pu.to_dev(stb), pu.to_dev(imb))
        torch.cuda.synchronize()
        for h in hooks:
            h()
        return out, grads

    for k in range(int(fx3["meta/steps"])):
        pre = "s%d/" % k
        stb, imb = gu.batches(fx3, pre)
        tape = gu.noise_tape(fx3, pre)
        before = {key: copy.deepcopy(n.state_dict()) for key, n in zip(keys, (st.netG, st.netD_im, st.netD_st, st.netD_se))}
        st_before = copy.deepcopy(st) if k else None
        ref = train_step(st, stb, imb, noise=NoiseTape(tape))
        out, grads = one(tr, stb, imb, tape)
        rep = pu.compare_step(out, ref, grads, oc.cascade)
        print("step %d continuing: " % k + "  ".join("%s %.2e" % (n, rep[n]) for n in ("loss_rel", "gradl2_G", "gradl2_D_im", "gradl2_D_st", "gradl2_D_se")), flush=True)
        if k:
            fresh = pu.make_trainer(oc, before, "fp32")
            pu.sync_from_oracle(fresh, st_before)
            out2, grads2 = one(fresh, stb, imb, tape)
            rep2 = pu.compare_step(out2, ref, grads2, oc.cascade)
            print("step %d fresh     : " % k + "  ".join("%s %.2e" % (n, rep2[n]) for n in ("loss_rel", "gradl2_G", "gradl2_D_im", "gradl2_D_st", "gradl2_D_se")), flush=True)
            same = all(torch.equal(grads[key][n], grads2[key][n]) for key in keys for n in grads[key])
            print("step %d continuing == fresh bit for bit: %s" % (k, same))
            del fresh
        for key, gk in pu.NETKEYS:
            if rep["gradl2_" + key] < 1e-3:
                continue
            print("  per tensor, %s:" % key)
            for name, g in ref[gk].items():
                gp, go = grads[key][name].double(), g.double()
                den = float((go * go).sum())
                if den == 0:
                    continue
                print("    %-44s |g| %.3e  rel %.2e  proj %.6f" % (name, den ** 0.5, float((gp - go).norm()) / den ** 0.5, float((gp * go).sum()) / den))
        pu.sync_from_oracle(tr, st)


if __name__ == "__main__":
    main()
