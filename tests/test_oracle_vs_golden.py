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


def _loaded_state(fx):
    cfg = gu.cfg_of(fx)
    st = make_state(cfg)
    for tag, net in (("G", st.netG), ("D_im", st.netD_im), ("D_st", st.netD_st), ("D_se", st.netD_se)):
        missing = net.load_state_dict(gu.group(fx, "before/" + tag), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    return cfg, st


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_state_dict_keys_match_reference(tag):
    fx = gu.load("step_%s.npz" % tag)
    _loaded_state(fx)  # strict load == identical key set and shapes


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_full_step_matches_reference(tag):
    fx = gu.load("step_%s.npz" % tag)
    torch.set_num_threads(int(fx["meta/seeds"][3]))
    cfg, st = _loaded_state(fx)
    stb, imb = gu.batches(fx)
    out = train_step(st, stb, imb, noise=NoiseTape(gu.noise_tape(fx)))
    # every scalar the reference logged
    for k in fx.files:
        if k.startswith("scalar/"):
            name = k.split("/", 1)[1]
            assert out[name] == pytest.approx(float(fx[k]), rel=1e-4, abs=1e-6), name
    # every gradient
    for tagn, key in (("D_se", "grads_D_se"), ("D_im", "grads_D_im"), ("D_st", "grads_D_st"), ("G", "grads_G")):
        ref = gu.group(fx, "grad/" + tagn)
        assert set(ref) == set(out[key])
        # biases in front of a BatchNorm have an exactly-zero true gradient (pure round-off in both
        # runs), so the floor is set by the network-wide gradient scale
        floor = 1e-5 * max(g.abs().max().item() for g in ref.values())
        for name, g in ref.items():
            err = (out[key][name].double() - g.double()).abs().max().item()
            assert err <= GRAD_TOL * g.abs().max().item() + floor, (tagn, name, err)
    # post-Adam parameters, BN running stats, SN u/v (summaries). The first Adam step moves every
    # parameter by lr*sign(g); where the true gradient is 0 (biases feeding a BatchNorm) the sign is
    # round-off noise, so parameters are held to 2*lr per element and buffers to fp32 tolerance.
    for tagn, net, lr in (("G", st.netG, cfg.g_lr), ("D_im", st.netD_im, cfg.d_lr),
                          ("D_st", st.netD_st, cfg.d_lr), ("D_se", st.netD_se, cfg.d_lr)):
        params = {k for k, _ in net.named_parameters()}
        for name, v in net.state_dict().items():
            ref = fx["after/%s/%s" % (tagn, name)]
            got = gu.summarise(v)
            if name in params:
                assert np.allclose(got[3:], ref[3:], rtol=0, atol=2.2 * lr), (tagn, name)
                assert abs(got[0] - ref[0]) <= 2.2 * lr * ref[2], (tagn, name)
            else:
                scale = max(ref[1] / max(ref[2], 1), 1e-6)
                assert np.allclose(got[3:], ref[3:], rtol=1e-3, atol=1e-3 * scale), (tagn, name)
                assert abs(got[0] - ref[0]) <= 1e-3 * ref[1] + 1e-6, (tagn, name)


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
