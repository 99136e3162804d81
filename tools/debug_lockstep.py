#!/usr/bin/env python3
"""Lock-step multi-step parity with the no-grad pass outputs compared per step (debugging aid)."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "cpcstoryvisualization-pytorch_amd"))
import torch
from tests import golden_util as gu, parity_util as pu
from oracle.cpcsv_oracle import NoiseTape, train_step
from cpcsv import runtime
tag = "plain"
fx3 = gu.load("steps3_%s.npz" % tag)
fx = gu.load(str(fx3["meta/weights_from"]))
oc, st, sds = pu.oracle_state_for(fx, gu.cfg_of(fx3))
runtime.set_deterministic(True)
tr = pu.make_trainer(oc, sds, "fp32")
for k in range(int(fx3["meta/steps"])):
    pre = "s%d/" % k
    stb, imb = gu.batches(fx3, pre)
    tape = gu.noise_tape(fx3, pre)
    # oracle's own no-grad outputs on ITS current state (before its step)
    import copy
    with torch.no_grad():
        td = oc.text_dim
        nt = NoiseTape(tape)
        st_motion = torch.cat((stb["description"][:, :, :td], stb["labels"]), 2)
        im_motion = torch.cat((imb["description"][:, :td], imb["labels"]), 1)
        G0 = copy.deepcopy(st.netG)
        _, o_st, _, _, o_cmu, _, _ = G0.sample_videos(st_motion, stb["description"][:, :, :td], noise=nt)
        _, o_im, _, _, o_imu, _, o_se = G0.sample_images(im_motion, imb["content"][:, :, :td], seg=True, noise=nt)
    ref = train_step(st, stb, imb, noise=NoiseTape(tape))
    pu.set_noise(tr.nets[0], pu.TapeSource(tape))
    seen = {}
    orig = tr._nograd_fakes
    def spy(*a, _o=orig):
        r = _o(*a)
        seen.update(zip(("st_fake", "c_mu", "im_fake", "cim_mu", "se_fake"), r))
        return r
    tr._nograd_fakes = spy
    grads = {}
    hooks = pu._capture_grads(tr, grads)
    out = tr.train_step(pu.to_dev(stb), pu.to_dev(imb))
    torch.cuda.synchronize()
    for h in hooks:
        h()
    tr._nograd_fakes = orig
    rep = pu.compare_step(out, ref, grads, oc.cascade)
    print("step", k, "nograd: st_fake %.2e im_fake %.2e se_fake %.2e c_mu %.2e | loss_rel %.2e gradl2 G %.2e D_im %.2e D_st %.2e D_se %.2e" % (
        pu.max_rel(seen["st_fake"].contiguous(), o_st.contiguous()), pu.max_rel(seen["im_fake"], o_im), pu.max_rel(seen["se_fake"], o_se),
        pu.max_rel(seen["c_mu"], o_cmu), rep["loss_rel"], rep["gradl2_G"], rep["gradl2_D_im"], rep["gradl2_D_st"], rep["gradl2_D_se"]), flush=True)
    pu.sync_from_oracle(tr, st)
