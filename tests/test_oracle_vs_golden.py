"""Pins the oracle: the CPU restatement must reproduce what the REAL reference produced
(fixtures written by oracle/gen_golden.py, which imports /root/reference)."""
import numpy as np
import pytest
import torch

from oracle.cpcsv_oracle import (NoiseTape, critic_loss, kl_term, make_state, multilabel_hit_rate,
                                 train_step)
from oracle.cpcsv_oracle.nets import dynamic_filter_1d
from tests import golden_util as gu

FWD_TOL = 2e-5    # fp32, different op fusion/summation order only
GRAD_TOL = 5e-4   # BN + spectral norm amplify
# CLEVR dims run ST=2 (batch=1 is impossible): BatchNorm1d over TWO rows has x_hat = +-1 and an inverse
# std of 2/|a-b|, so fp32 round-off in the story branch is amplified ~10x more than at ST=3
# seq: the generator's gradient through the order critic's MSE term passes BatchNorm1d over ST=3 rows and BatchNorm3d over 12
# values: 1.5e-3 relative L2 oracle-vs-reference (critics, incl. the order critic itself: 5e-7)
GRAD_TOL_OF = {"clevr": 1e-2, "seq": 1e-2}


def _loaded_state(fx):
    cfg = gu.cfg_of(fx)
    st = make_state(cfg)
    sds = gu.state_dicts(fx)
    for tag, net in (("G", st.netG), ("D_im", st.netD_im), ("D_st", st.netD_st), ("D_se", st.netD_se)):
        missing = net.load_state_dict(sds[tag], strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    return cfg, st


# clevr = BASELINE config 1 dims (T=4, text 72, labels 15, ST=2/IM=8); seq = USE_SEQ_CONSISTENCY (VideoEncoder order critic
# + create_random_shuffle)
TAGS = ["plain", "cascade", "clevr", "seq"]


@pytest.mark.parametrize("tag", TAGS)
def test_state_dict_keys_match_reference(tag):
    fx = gu.load("step_%s.npz" % tag)
    _loaded_state(fx)  # strict load == identical key set and shapes


@pytest.mark.parametrize("tag", TAGS)
def test_full_step_matches_reference(tag):
    fx = gu.load("step_%s.npz" % tag)
    torch.set_num_threads(int(fx["meta/seeds"][3]))
    cfg, st = _loaded_state(fx)
    stb, imb = gu.batches(fx)
    out = train_step(st, stb, imb, noise=NoiseTape(gu.noise_tape(fx)), shuffle=gu.shuffle_plan_of(fx))
    # every scalar the reference logged
    for k in fx.files:
        if k.startswith("scalar/"):
            name = k.split("/", 1)[1]
            assert out[name] == pytest.approx(float(fx[k]), rel=1e-4, abs=1e-6), name
    # every gradient
    for tagn, key in (("D_se", "grads_D_se"), ("D_im", "grads_D_im"), ("D_st", "grads_D_st"), ("G", "grads_G")):
        ref = gu.group(fx, "grad/" + tagn)
        big = {k[len("gradsum/%s/" % tagn):] for k in fx.files if k.startswith("gradsum/%s/" % tagn)}   # summaries only (order critic)
        assert set(ref) | big == set(out[key])
        if big:
            e_abs, e_head = gu.grad_summary_error(fx, "gradsum/" + tagn, {k: out[key][k] for k in big})
            assert e_abs < 1e-3 and e_head < 1e-3, (tagn, e_abs, e_head)
        # biases in front of a BatchNorm have an exactly-zero true gradient (pure round-off in both
        # runs), so the floor is set by the network-wide gradient scale
        floor = 1e-5 * max(g.abs().max().item() for g in ref.values())
        for name, g in ref.items():
            err = (out[key][name].double() - g.double()).abs().max().item()
            assert err <= GRAD_TOL_OF.get(tag, GRAD_TOL) * g.abs().max().item() + floor, (tagn, name, err)
    # post-Adam parameters, BN running stats, SN u/v (summaries)
    for tagn, net, lr in (("G", st.netG, cfg.g_lr), ("D_im", st.netD_im, cfg.d_lr),
                          ("D_st", st.netD_st, cfg.d_lr), ("D_se", st.netD_se, cfg.d_lr)):
        gu.check_after_state(fx, "after/" + tagn, net, lr, steps=1, buf_rtol=1e-2 if tag == "clevr" else 1e-3)


# Free-running K-step comparisons are limited by Adam, not by the kernels: in the first steps Adam's update is
# lr*g/(|g|+1e-8) ~ lr*sign(g), so a gradient entry whose TRUE value is ~0 (biases in front of a BatchNorm; the columns
# of the critics' head conv that multiply conditioning inputs which are zero for the whole batch - CA_NET's ReLU zeroes
# many c_mu entries, model.py:48-50 - where only the ~1e-9 spectral-norm rank-1 term survives) moves its weight by up
# to +-lr on the sign of round-off. Two runs of the REAL reference with different thread counts diverge the same way.
# Measured here (oracle vs reference, this fixture): gradient L2 error 4e-6 / 2e-4 / 2.5e-2 and losses 3e-7 / 7e-5 /
# 2e-4 at steps 0 / 1 / 2. Bounds below are ~5x that.
STEP_LOSS_TOL = (1e-4, 5e-4, 2e-3)
STEP_GRAD_TOL = (1e-3, 2e-2, 1.5e-1)


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_three_steps_match_reference(tag):
    """K=3 consecutive steps of the real reference (fresh batch and noise each step): state carry across steps
    (Adam moments, SN u/v advancing in the no-grad pass too, BN running statistics updated twice per step)."""
    from oracle import conditioning as COND
    from tests import parity_util as pu
    fx3 = gu.load("steps3_%s.npz" % tag)
    fx = gu.load(str(fx3["meta/weights_from"]))
    torch.set_num_threads(int(fx3["meta/seeds"][3]))
    cfg, st = _loaded_state(fx)
    cfg = gu.cfg_of(fx3)                      # (the three-step fixtures have their own batch sizes)
    st.cfg = cfg
    nets = (("D_se", "grads_D_se"), ("D_im", "grads_D_im"), ("D_st", "grads_D_st"), ("G", "grads_G"))
    for k in range(int(fx3["meta/steps"])):
        pre = "s%d/" % k
        stb, imb = gu.batches(fx3, pre)
        tape = gu.noise_tape(fx3, pre)
        snap = pu.oracle_snapshot(st)
        out = train_step(st, stb, imb, noise=NoiseTape(tape))
        err = lambda o: sum(sum(gu.grad_summary_error(fx3, pre + "gradsum/" + tagn, o[gk])) for tagn, gk in nets)
        tol_of = lambda tagn: STEP_GRAD_TOL[k]
        if any(max(gu.grad_summary_error(fx3, pre + "gradsum/" + tagn, out[gk])) >= tol_of(tagn) for tagn, gk in nets):
            # the reference and its restatement are two fp32 evaluations with different summation orders: a pre-activation within
            # a few round-offs of its kink may sit on the other side in the recorded run - resolved, not tolerated (the same
            # mechanism the GPU lock-step tests use, oracle/conditioning.py): the oracle must match the record for ONE assignment
            # of sides to the elements it lists as near a kink, within the same bounds
            out, st, kept = COND.match_kink_sides(cfg, snap, stb, imb, tape, err)
            print("step %d: near-kink elements resolved against the reference's record: %s" % (k, kept))
        for key in fx3.files:
            if key.startswith(pre + "scalar/"):
                name = key.split("/", 2)[2]
                tol = 0.35 if name.endswith("_acc") else STEP_LOSS_TOL[k]      # accuracies are counts over 4-9 labels
                assert out[name] == pytest.approx(float(fx3[key]), rel=tol, abs=1e-6), (k, name)
        for tagn, gk in (("D_se", "grads_D_se"), ("D_im", "grads_D_im"), ("D_st", "grads_D_st"), ("G", "grads_G")):
            e_abs, e_head = gu.grad_summary_error(fx3, pre + "gradsum/" + tagn, out[gk])
            assert e_abs < tol_of(tagn) and e_head < tol_of(tagn), (k, tagn, e_abs, e_head)
        for tagn, net, lr in (("G", st.netG, cfg.g_lr), ("D_im", st.netD_im, cfg.d_lr),
                              ("D_st", st.netD_st, cfg.d_lr), ("D_se", st.netD_se, cfg.d_lr)):
            gu.check_after_state(fx3, pre + "after/" + tagn, net, lr, steps=k + 1, buf_rtol=(1e-3, 5e-3, 2e-2)[k])


def test_nograd_forward_matches_reference():
    fx = gu.load("step_plain.npz")
    cfg, st = _loaded_state(fx)
    stb, imb = gu.batches(fx)
    td = cfg.text_dim
    tape = NoiseTape(gu.noise_tape(fx))
    st_motion = torch.cat((stb["description"][:, :, :td], stb["labels"]), 2)
    im_motion = torch.cat((imb["description"][:, :td], imb["labels"]), 1)
    with torch.no_grad():
        _, v, _, _, c_mu, c_lv, _ = st.netG.sample_videos(st_motion, stb["description"][:, :, :td], noise=tape)
        _, im, _, _, cim_mu, cim_lv, se = st.netG.sample_images(im_motion, imb["content"][:, :, :td], seg=True, noise=tape)
    assert not v.is_contiguous() and tuple(v.shape) == (cfg.st_batch, 3, cfg.video_len, 64, 64)
    for got, name in ((v, "st_fake"), (im, "im_fake"), (se, "se_fake"), (c_mu, "c_mu"), (c_lv, "c_logvar"),
                      (cim_mu, "cim_mu"), (cim_lv, "cim_logvar")):
        assert gu.rel_err(got, fx["nograd/" + name]) < FWD_TOL, name


def test_micro_ops_match_reference():
    fx = gu.load("ops.npz")
    sig = torch.from_numpy(fx["dfl/sig"]).requires_grad_()
    taps = torch.from_numpy(fx["dfl/taps"]).requires_grad_()
    out = dynamic_filter_1d(sig, taps, 10)
    assert gu.rel_err(out, fx["dfl/out"]) < 1e-6
    out.backward(torch.from_numpy(fx["dfl/up"]))
    assert gu.rel_err(sig.grad, fx["dfl/dsig"]) < 1e-6
    assert gu.rel_err(taps.grad, fx["dfl/dtaps"]) < 1e-6
    kl = kl_term(torch.from_numpy(fx["kl/mu"]), torch.from_numpy(fx["kl/logvar"]))
    assert kl.item() == pytest.approx(float(fx["kl/out"]), rel=1e-6)
    acc = multilabel_hit_rate(torch.from_numpy(fx["acc/logits"]), torch.from_numpy(fx["acc/labels"]))
    assert acc == pytest.approx(float(fx["acc/out"]), rel=1e-12)


def test_quirks():
    """Appendix A items that change results: tiled c_mu rows, min batch 2."""
    fx = gu.load("step_plain.npz")
    cfg, st = _loaded_state(fx)
    stb, _ = gu.batches(fx)
    td = cfg.text_dim
    st_motion = torch.cat((stb["description"][:, :, :td], stb["labels"]), 2)
    with pytest.raises(ValueError):   # BatchNorm1d on a single row, model.py:306-308
        st.netG.sample_videos(st_motion[:1], stb["description"][:1, :, :td])


EVAL_TOL = 2e-5   # forward only, fp32, same weights: summation order


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_eval_mode_matches_reference(tag):
    """The oracle's EVAL-mode forward (reference inference.py:88-89: netG.eval() under no_grad; BatchNorm running statistics,
    spectral norm frozen at the stored u / v) against tensors the imported reference produced (oracle/gen_golden.py
    reference_eval) on the state twelve training steps of the reference leave behind: generator outputs of both sampling calls, the three
    critics' features, conditional logits and category logits - and u / v / running statistics must not move."""
    fx = gu.load("eval_%s.npz" % tag)
    torch.set_num_threads(int(fx["meta/seeds"][3]))
    cfg = gu.cfg_of(fx)
    st = make_state(cfg)
    nets = (("G", st.netG), ("D_im", st.netD_im), ("D_st", st.netD_st), ("D_se", st.netD_se))
    for name, net in nets:
        res = net.load_state_dict(gu.state_dicts(fx, "state")[name], strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        net.eval()
    stb, imb = gu.batches(fx)
    td = cfg.text_dim
    st_motion = torch.cat((stb["description"][:, :, :td], stb["labels"]), 2)
    im_motion = torch.cat((imb["description"][:, :td], imb["labels"]), 1)
    noise = NoiseTape(gu.noise_tape(fx))
    ref = gu.group(fx, "eval")
    with torch.no_grad():
        _, sv, _, _, c_mu, c_lv, sseg = st.netG.sample_videos(st_motion, stb["description"][:, :, :td], seg=True, noise=noise)
        _, si, _, _, i_mu, i_lv, iseg = st.netG.sample_images(im_motion, imb["content"][:, :, :td], seg=True, noise=noise)
        got = {"st_fake": sv.contiguous(), "st_seg": sseg.contiguous(), "im_fake": si, "se_fake": iseg, "c_mu": c_mu, "c_logvar": c_lv,
               "cim_mu": i_mu, "cim_logvar": i_lv}
        for name, net, imgs, cond in (("D_im", st.netD_im, imb["images"], ref["im_cond"]), ("D_se", st.netD_se, imb["images_seg"], ref["im_cond"]),
                                      ("D_st", st.netD_st, stb["images"], ref["st_cond"])):
            feats = net(imgs)
            got[name + "_feats"] = feats
            got[name + "_logits"] = net.get_cond_logits(feats, cond)
            if net.cate_classify is not None:
                got[name + "_cate"] = net.cate_classify(feats)
    for k, v in got.items():
        assert v.shape == ref[k].shape, (k, v.shape, ref[k].shape)
        assert gu.rel_err(v, ref[k]) < EVAL_TOL, (k, gu.rel_err(v, ref[k]))
    assert {k for k in ref if k.endswith(("_feats", "_logits", "_cate"))} <= set(got)
    # nothing moved: every buffer and parameter summary after the eval calls equals the stored state's
    for name, net in nets:
        for k, v in net.state_dict().items():
            key = "after_eval/%s/%s" % (name, k)
            assert np.allclose(gu.summarise(v), fx[key], rtol=1e-6, atol=1e-7), key


@pytest.mark.parametrize("name", ["fullwidth_plain", "fullwidth_cascade", "fullwidth_clevr", "fullwidth_bench"])
def test_fullwidth_fixture_weights_come_from_the_seed(name):
    """The full-width reference records (oracle/gen_golden.py --fullwidth) hold no weights: the imported reference built under
    torch.manual_seed(seed) and oracle.make_state(cfg, seed) produce the same 158 M values (asserted tensor by tensor when the
    fixture is made). Here, without the reference: make_state(cfg, seed) reproduces the recorded checksum of those weights, and
    synthetic_batch(cfg, seed) the recorded batch (the bench record keeps checksums of it only)."""
    from oracle.cpcsv_oracle import synthetic_batch
    fx = gu.load(name + ".npz")
    cfg = gu.cfg_of(fx)
    seed_w, seed_data = int(fx["meta/seeds"][0]), int(fx["meta/seeds"][1])
    st = make_state(cfg, seed=seed_w)
    total = sum(float(v.double().abs().sum()) for n in (st.netG, st.netD_im, st.netD_st, st.netD_se) for v in n.state_dict().values())
    assert abs(total - float(fx["meta/weights_sum"])) <= 1e-9 * total, (total, float(fx["meta/weights_sum"]))
    stb, imb = synthetic_batch(cfg, seed=seed_data)
    for tag, batch in (("st", stb), ("im", imb)):
        for k, v in batch.items():
            if "batch/%s/%s" % (tag, k) in fx.files:
                assert np.array_equal(v.numpy(), fx["batch/%s/%s" % (tag, k)]), (tag, k)
            else:
                assert np.array_equal(gu.summarise(v), fx["batchsum/%s/%s" % (tag, k)]), (tag, k)
