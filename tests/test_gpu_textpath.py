"""The fused text / motion encoder path (cpcsv/textpath.py, csrc/text.hip: ~10 stage launches forward, ~11 backward) against the
per-layer path it replaces (reference model.py:37-65,302-346,371-378; layers.py:69-80): same weights, same inputs, same recorded
noise -> the joint code, mu / logvar of both calls, every parameter gradient of the nine small layers and their BatchNorms, the
BatchNorm running statistics. Both are this library's fp32 arithmetic; they differ in summation order only (tolerances below).
The comparison with the ORACLE (and through it the reference) is tests/test_gpu_step.py, which runs the fused path by default."""
import copy

import pytest
import torch

from oracle.cpcsv_oracle.config import clevr_cfg, pororo_cfg, tiny_cfg
from tests.parity_util import TapeSource, apply_cfg, set_noise

pytestmark = pytest.mark.gpu


class _Recorder:
    def __init__(self):
        self.tape = []

    def __call__(self, shape):
        t = torch.randn(tuple(shape), device="cuda")
        self.tape.append(t)
        return t


def _gen(oc):
    apply_cfg(oc)
    import model as mod
    from miscc.utils import weights_init
    g = mod.StoryGAN(oc.video_len)
    g.apply(weights_init)
    with torch.no_grad():                     # BatchNorm gains / biases off their init values so that every gradient path is live
        for m in (g.m_net[1], g.c_net[1], g.image_net[1], g.filter_net[1]):
            m.weight.add_(0.3 * torch.randn_like(m.weight))
            m.bias.add_(0.2 * torch.randn_like(m.bias))
        for lin in (g.ca_net.fc, g.m_net[0], g.c_net[0], g.image_net[0], g.filter_net[0]):
            lin.weight.mul_(8.0)               # N(0, .02) initial weights give near-constant pre-activations
            lin.bias.add_(0.1 * torch.randn_like(lin.bias))
    return g.cuda().train()


class _Stop(Exception):
    pass


def _text_params(g):
    from cpcsv import textpath as TP
    names = ["ca.w", "ca.b", "m.w", "m.b", "m.g", "m.be", "c.w", "c.b", "c.g", "c.be", "ih_m.w", "ih_m.b", "hh_m.w", "hh_m.b",
             "ih_c.w", "ih_c.b", "hh_c.w", "hh_c.b", "i.w", "i.b", "i.g", "i.be", "f.w", "f.b", "f.g", "f.be"]
    return list(zip(names, TP._params(g)))


def _bn_state(g):
    out = {}
    for nm, m in (("m", g.m_net[1]), ("c", g.c_net[1]), ("i", g.image_net[1]), ("f", g.filter_net[1])):
        out[nm + ".rm"], out[nm + ".rv"] = m.running_mean.clone(), m.running_var.clone()
    return out


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item(), ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("cfgname,st,im", [("tiny", 3, 4), ("clevr", 2, 8), ("pororo", 12, 60), ("pororo", 2, 64)])
def test_fused_text_path_matches_the_per_layer_path(cfgname, st, im):
    from cpcsv import runtime
    runtime.set_compute_dtype("fp32")
    runtime.set_deterministic(True)
    try:
        mk = {"tiny": tiny_cfg, "clevr": clevr_cfg, "pororo": pororo_cfg}[cfgname]
        oc = mk(gf_dim=4, gf_seg_dim=16, df_dim=8)
        torch.manual_seed(7)
        g = _gen(oc)
        t, md, td = oc.video_len, oc.text_dim + oc.label_num, oc.text_dim
        ins = (torch.randn(st, t, md, device="cuda"), torch.randn(st, t, td, device="cuda"),
               torch.randn(im, md, device="cuda"), torch.randn(im, t, td, device="cuda"))
        state0 = copy.deepcopy(g.state_dict())
        rec = _Recorder()
        report, worst = [], 0.0
        res = {}
        for fused in (False, True):
            g.load_state_dict(state0)
            src = rec if not fused else TapeSource(rec.tape)
            captured = {}

            # run the pass up to the joint code, then a fixed random linear functional of (zmc, mu, logvar) as the loss
            from cpcsv import textpath as TP
            TP.ENABLED = fused
            set_noise(g, src)
            for p in g.parameters():
                p.grad = None
            bs = ins[0].shape[0]
            st_flat = ins[1].reshape(-1, t * td)
            im_flat = ins[3].reshape(-1, t * td)
            temp = ins[0].reshape(-1, md)
            orig = g._decode_both

            def grab(zmc_all, nst, nim, bs_, vl, seg, temp_, im_m, r_mu, r_lv, c_mu, c_lv):
                captured.update(zmc=zmc_all, r_mu=r_mu, r_lv=r_lv, c_mu=c_mu, c_lv=c_lv)
                raise _Stop()
            g._decode_both = grab
            try:
                g._sample_both(ins[0], ins[1], ins[2], ins[3], True, bs, t, st_flat, temp, im_flat)
            except _Stop:
                pass
            finally:
                g._decode_both = orig
                TP.ENABLED = True
            if "w" not in res:
                gen = torch.Generator(device="cuda").manual_seed(11)
                res["w"] = {k: torch.randn(v.shape, device="cuda", generator=gen) for k, v in captured.items()}
            loss = sum((captured[k].float() * res["w"][k]).sum() for k in captured)
            loss.backward()
            torch.cuda.synchronize()
            res[fused] = dict(out={k: v.detach().float().clone() for k, v in captured.items()},
                              grads={n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in _text_params(g)},
                              bn=_bn_state(g))
        a, b = res[True], res[False]
        # forward: BIT-identical (same products in the same order, the same 16-row statistics partials combined in double)
        for k in b["out"]:
            same = torch.equal(a["out"][k], b["out"][k])
            e2, em = _rel(a["out"][k], b["out"][k])
            report.append("out  %-8s %s  (L2 %.2e max %.2e)" % (k, "bit-identical" if same else "DIFFERS", e2, em))
            worst = max(worst, 0.0 if same else 1e9)
        for k in b["bn"]:
            same = torch.equal(a["bn"][k], b["bn"][k])
            e2, em = _rel(a["bn"][k], b["bn"][k])
            report.append("bn   %-8s %s  (L2 %.2e max %.2e)" % (k, "bit-identical" if same else "DIFFERS", e2, em))
            worst = max(worst, 0.0 if same else 1e9)
        # backward: another summation order (column-owned BatchNorm sums instead of atomics, one weight-gradient launch)
        for n in b["grads"]:
            if b["grads"][n] is None or a["grads"][n] is None:
                report.append("grad %-8s MISSING (fused: %s, per layer: %s)" % (n, a["grads"][n] is not None, b["grads"][n] is not None))
                worst = max(worst, 1e9)
                continue
            if n in ("m.b", "c.b", "i.b", "f.b"):
                # a bias in front of a train-mode BatchNorm: its true gradient is exactly 0, both sides hold round-off of the size
                # eps * |weight gradient|; compared on that scale
                wscale = b["grads"][n[0] + ".w"].abs().max().item()
                err = (a["grads"][n] - b["grads"][n]).abs().max().item() / (wscale + 1e-30)
                report.append("grad %-8s |a-b| / max|dW| %.2e   (|g| %.3e: true value 0)" % (n, err, b["grads"][n].norm().item()))
                worst = max(worst, err / 1e-5)
                continue
            e2, em = _rel(a["grads"][n], b["grads"][n])
            report.append("grad %-8s L2 %.2e max %.2e   |g| %.3e" % (n, e2, em, b["grads"][n].norm().item()))
            worst = max(worst, e2 / 2e-5)
        print("\n".join(report))
        assert worst <= 1.0, "\n" + "\n".join(report)
    finally:
        runtime.set_deterministic(False)
        runtime.set_compute_dtype("bf16")


def test_fused_text_path_in_bf16_mode_feeds_the_decoder():
    """bf16 compute dtype: the joint code leaves the fused path as a bf16 matrix and its gradient comes back as one."""
    from cpcsv import runtime
    runtime.set_compute_dtype("bf16")
    oc = tiny_cfg()
    torch.manual_seed(3)
    g = _gen(oc)
    t, md, td = oc.video_len, oc.text_dim + oc.label_num, oc.text_dim
    st, im = 3, 4
    (_, st_fake, _, _, c_mu, c_lv, _), (_, im_fake, _, _, cim_mu, cim_lv, se) = g.sample_both(
        torch.randn(st, t, md, device="cuda"), torch.randn(st, t, td, device="cuda"), torch.randn(im, md, device="cuda"),
        torch.randn(im, t, td, device="cuda"), seg=True)
    (st_fake.sum() + im_fake.sum() + se.sum() + c_mu.sum() + cim_lv.sum()).backward()
    torch.cuda.synchronize()
    for n, p in _text_params(g):
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert torch.isfinite(st_fake).all() and torch.isfinite(im_fake).all()
