#!/usr/bin/env python3
"""Where do two arms of the product (default: fused text path on / off) part ways in the lock-step run of the steps3 fixture
(tests/parity_util.run_multistep_parity)? Per step and critic: checksums of what compute_discriminator_loss receives (real, fake,
condition), its loss terms, and the critic's gradient norm right before its optimiser step; per step the checksums of the
no-grad pass's outputs.   python tools/lockstep_diag.py [ENV=VAL ...]   (the arm B environment; default CPCSV_TEXT_FUSED=0)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
if "torch" not in sys.modules:
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
import torch  # noqa: E402


def cs(t):
    t = t.detach().double()
    return "%.12e/%.12e" % (float(t.sum()), float(t.abs().sum()))


def run(arm):
    from cpcsv import runtime, textpath
    from tests import golden_util as gu
    from tests import parity_util as pu
    import miscc.utils as MU
    import trainer as T
    textpath.ENABLED = arm == "A"
    fx3 = gu.load("steps3_plain.npz")
    fx = gu.load(str(fx3["meta/weights_from"]))
    oc, st, sds = pu.oracle_state_for(fx, gu.cfg_of(fx3))
    runtime.set_deterministic(True)
    tr = pu.make_trainer(oc, sds, "fp32")
    rows = []
    orig = T.compute_discriminator_loss

    def spy(netD, real, fake, rl, fl, cate, cond, gpus, **kw):
        out = orig(netD, real, fake, rl, fl, cate, cond, gpus, **kw)
        torch.cuda.synchronize()
        rows.append("   D %-14s real %s fake %s cond %s | errD %.9e real %.9e wrong %.9e fake %.9e"
                    % (type(netD).__name__, cs(real), cs(fake), cs(cond), float(out[0]), float(out[1]), float(out[2]), float(out[3])))
        return out
    T.compute_discriminator_loss = spy
    from oracle.cpcsv_oracle import NoiseTape, train_step
    try:
        for k in range(int(fx3["meta/steps"])):
            pre = "s%d/" % k
            stb, imb = gu.batches(fx3, pre)
            tape = gu.noise_tape(fx3, pre)
            train_step(st, stb, imb, noise=NoiseTape(tape))
            pu.set_noise(tr.nets[0], pu.TapeSource(tape))
            grads = {}
            hooks = pu._capture_grads(tr, grads)
            rows.append("step %d" % k)
            out = tr.train_step(pu.to_dev(stb), pu.to_dev(imb))
            torch.cuda.synchronize()
            for h in hooks:
                h()
            for key in ("D_se", "D_im", "D_st", "G"):
                n2 = sum(float((g.double() ** 2).sum()) for g in grads[key].values()) ** 0.5
                rows.append("   |grad %s| %.12e" % (key, n2))
            rows.append("   losses " + " ".join("%s=%.9e" % (kk, float(v)) for kk, v in sorted(out.items()) if "Acc" not in kk))
            pu.sync_from_oracle(tr, st)
    finally:
        T.compute_discriminator_loss = orig
        runtime.set_deterministic(False)
    return rows


if __name__ == "__main__":
    a = run("A")
    b = run("B")
    for x, y in zip(a, b):
        mark = "   " if x == y else "!! "
        print(mark + "A " + x)
        if x != y:
            print(mark + "B " + y)
