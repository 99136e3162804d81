"""Shared helpers of the GPU parity tests and __graft_entry__.smoke(): build the PRODUCT nets
(HIP path) from an OracleCfg, load a golden fixture / oracle weights, run a step on both sides."""
import copy
import types

import numpy as np
import torch

from tests import golden_util as gu


def apply_cfg(oc):
    """Push an OracleCfg into the product's global cfg (miscc.config.cfg)."""
    from miscc.config import cfg
    cfg.VIDEO_LEN = oc.video_len
    cfg.LABEL_NUM = oc.label_num
    cfg.TEXT.DIMENSION = oc.text_dim
    cfg.GAN.CONDITION_DIM = oc.cond_dim
    cfg.GAN.Z_DIM = oc.z_dim
    cfg.GAN.DF_DIM = oc.df_dim
    cfg.GAN.GF_DIM = oc.gf_dim
    cfg.GAN.GF_SEG_DIM = oc.gf_seg_dim
    cfg.SEGMENT_LEARNING = oc.segment_learning
    cfg.SEGMENT_RATIO = oc.segment_ratio
    cfg.IMAGE_RATIO = oc.image_ratio
    cfg.RECONSTRUCT_LOSS = oc.reconstruct_loss
    cfg.CASCADE_MODEL = oc.cascade
    cfg.USE_SEQ_CONSISTENCY = getattr(oc, "use_seq_consistency", False)
    cfg.CONSISTENCY_RATIO = getattr(oc, "consistency_ratio", 1.0)
    cfg.EVALUATE_FID_SCORE = False
    cfg.TRAIN.COEFF.KL = oc.kl_coeff
    cfg.TRAIN.IM_BATCH_SIZE = oc.im_batch
    cfg.TRAIN.ST_BATCH_SIZE = oc.st_batch
    cfg.TRAIN.GENERATOR_LR = oc.g_lr
    cfg.TRAIN.DISCRIMINATOR_LR = oc.d_lr
    cfg.GPU_ID = '0'
    return cfg


def product_nets(oc):
    apply_cfg(oc)
    if oc.cascade:
        import cascade_model as mod
    else:
        import model as mod
    from miscc.utils import weights_init
    nets = [mod.StoryGAN(oc.video_len), mod.STAGE1_D_IMG(), mod.STAGE1_D_STY_V2(), mod.STAGE1_D_SEG()]
    for n in nets:
        n.apply(weights_init)
    return nets


def make_trainer(oc, state_dicts, dtype="fp32", device="cuda"):
    """GANTrainer with product nets carrying the given state dicts (dict name -> state_dict)."""
    from cpcsv import runtime
    import trainer as T
    runtime.set_compute_dtype(dtype)
    nets = product_nets(oc)
    for n, key in zip(nets, ("G", "D_im", "D_st", "D_se")):
        n.load_state_dict(state_dicts[key], strict=True)
        n.to(device)
    tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
    tr.setup(tuple(nets))
    return tr


def to_dev(batch, device="cuda"):
    return {k: v.to(device) for k, v in batch.items()}


def grads_of(net, opt=None, wire_of=None):
    """name -> gradient in master layout. Weights on the deferred-update path have no materialised .grad: the optimiser
    rebuilds it from the layer's accumulator (cpcsv.optim.FusedAdam.export_grad) - or, behind a data-parallel exchange with the
    bf16 payload, from the bucket's wire buffer, where the reduced values live."""
    get = (lambda p: opt.export_grad(p, wire_of)) if opt is not None and hasattr(opt, "export_grad") else (lambda p: p.grad)
    return {k: get(p).detach().float().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}


class TapeSource:
    def __init__(self, tape):
        self.tape, self.pos = list(tape), 0

    def __call__(self, shape):
        t = self.tape[self.pos]
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape), self.pos)
        self.pos += 1
        return t


def set_noise(netG, source):
    netG.noise_source = source
    netG.ca_net.noise_source = source


def max_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def _force_two_stream_nograd(tr):
    """Give the trainer the list of generator layers that repack after an optimiser step (it normally learns it in
    its first step), so that ALREADY the first step runs the no-grad pass as two halves on two streams."""
    import copy
    from cpcsv import modules as M
    netG = tr.nets[0]
    keep = copy.deepcopy(netG.state_dict())
    for lay in [m for m in netG.modules() if hasattr(m, "note_batch")]:
        lay._pending = 0
    M.PACK_LOG = []
    try:
        # a throw-away pass on random inputs of the right widths just to see which layers pack
        from miscc.config import cfg
        st = torch.randn(2, cfg.VIDEO_LEN, cfg.TEXT.DIMENSION + cfg.LABEL_NUM, device="cuda")
        sc = torch.randn(2, cfg.VIDEO_LEN, cfg.TEXT.DIMENSION, device="cuda")
        with torch.no_grad():
            netG.sample_videos(st, sc)
            netG.sample_images(st[:, 0], sc, seg=True)
        tr._g_packs = list(M.PACK_LOG)
    finally:
        M.PACK_LOG = None
    netG.load_state_dict(keep)
    for lay in [m for m in netG.modules() if hasattr(m, "note_batch")]:
        lay._pending = 0
    assert tr._g_packs


LOSS_NAMES = {"G_loss": "G/loss", "im_D_loss": "img_D/loss", "st_D_loss": "st_D/loss", "se_D_loss": "seg_D/loss",
              "im_D_real": "img_D/real", "im_D_wrong": "img_D/wrong", "im_D_fake": "img_D/fake",
              "st_D_real": "st_D/real", "st_D_wrong": "st_D/wrong", "st_D_fake": "st_D/fake",
              "se_D_real": "seg_D/real", "se_D_wrong": "seg_D/wrong", "se_D_fake": "seg_D/fake",
              "im_KL": "G/im_KL", "st_KL": "G/st_KL", "im_G": "G/im", "st_G": "G/st", "se_G": "G/se"}
CASCADE_NAMES = {"video_latent": "G/video_vae_loss", "image_latent": "G/image_vae_loss", "reconstruct": "G/reconstruct_loss"}
ACC_NAMES = {"im_D_acc": "Accuracy/im_D", "se_D_acc": "Accuracy/se_D", "im_G_acc": "Accuracy/im_G",
             "se_G_acc": "Accuracy/se_G", "st_G_acc": "Accuracy/st_G"}
NETKEYS = (("G", "grads_G"), ("D_im", "grads_D_im"), ("D_st", "grads_D_st"), ("D_se", "grads_D_se"))

# (losses rel, whole-net gradient relative L2, per-element error / tensor max) of ONE step from identical state.
# fp32: exact-f32 MFMA, differences are summation order only. bf16: operands rounded to 8 bits of mantissa through
# ~40 layers at the fixture's 2-64 channel widths (the harshest case: no averaging over wide reductions): measured
# losses 0.3-0.5 %, 0.08-0.12 relative L2 on the critics' gradient vectors and 0.19-0.25 on the generator's, spread
# evenly over its layers (per-tensor split: profiles/r02_parity.txt). The small dense text/motion-encoder layers run in
# fp32 even in bf16 mode (cpcsv.modules.KernelLayer.compute_f32): with bf16 operands there the generator's error was
# 0.33-0.69, dominated by the layers behind BatchNorm1d over ST=3 rows.
STEP_TOL = {"fp32": (2e-4, 5e-3, 5e-2), "bf16": (3e-2, 0.35, 0.6)}


class oracle_threads:
    """The oracle runs with the CPU thread count the reference's own run used when the fixture was recorded (fixture meta/seeds[3]):
    its summation order - and with it which side of a LeakyReLU kink a round-off-sized activation lands on, and through Adam's
    lr * sign(g) first step the whole state after a step - depends on the thread count (tools/oracle_host_check.py: 8 / 32 / 256
    threads reach three different step-1 states on one host). The hosts' own default (256 on the MI355X boxes) is also 40x slower."""

    def __init__(self, fx):
        self.n = int(fx["meta/seeds"][3]) if "meta/seeds" in fx.files else None

    def __enter__(self):
        self.keep = torch.get_num_threads()
        if self.n:
            torch.set_num_threads(self.n)

    def __exit__(self, *a):
        torch.set_num_threads(self.keep)


def oracle_state_for(fx, oc=None):
    from oracle.cpcsv_oracle import make_state
    oc = oc or gu.cfg_of(fx)
    st = make_state(oc)
    sds = gu.state_dicts(fx)
    for key, net in (("G", st.netG), ("D_im", st.netD_im), ("D_st", st.netD_st), ("D_se", st.netD_se)):
        net.load_state_dict(sds[key])
    return oc, st, sds


def compare_step(out, ref, grads, cascade, seq=False):
    """Scalars, accuracies and gradients of one product step (out, grads) against the oracle's (ref)."""
    rep = {}
    names = dict(LOSS_NAMES)
    if cascade:
        names.update(CASCADE_NAMES)
    if seq:
        names.update({"st_D_consistency": "st_D/order", "st_G_consistency": "G/consistency"})
    worst, wname = 0.0, ""
    for rk, pk in names.items():
        got, want = float(out[pk]), float(ref[rk])
        e = abs(got - want) / (abs(want) + 1e-8)
        if e > worst:
            worst, wname = e, rk
    rep["loss_rel"], rep["worst_loss"] = worst, wname
    # the critics' losses are taken BEFORE any optimiser step of this train_step, the generator's after all three critic updates
    rel = lambda rk, pk: abs(float(out[pk]) - float(ref[rk])) / (abs(float(ref[rk])) + 1e-8)
    rep["loss_rel_D"] = max(rel(rk, pk) for rk, pk in names.items() if "_D_" in rk)
    rep["loss_rel_G"] = max(rel(rk, pk) for rk, pk in names.items() if "_D_" not in rk)
    rep["worst_losses"] = "; ".join("%s %.6g/%.6g" % (rk, float(out[pk]), float(ref[rk])) for rk, pk in names.items()
                                    if abs(float(out[pk]) - float(ref[rk])) > 1e-4 * (abs(float(ref[rk])) + 1e-8))
    # accuracies are hit counts / positive-label counts (miscc/utils.py:313-321): equal unless a logit sits on 0
    rep["acc_abs"] = max(abs(float(out[pk]) - float(ref[rk])) for rk, pk in ACC_NAMES.items())
    for key, gk in NETKEYS:
        refg = ref[gk]
        scale = max(g.abs().max().item() for g in refg.values())
        # two views of the error: per tensor, the max element error against that tensor's max (any indexing / tap /
        # border bug shows up as O(1)); per net, the relative L2 error of the WHOLE gradient vector. Why no tight
        # per-element bound: a BN output within round-off of 0 flips ONE LeakyReLU/ReLU mask, which moves one channel's
        # small-sample sums by ~1 % while everything else agrees to 1e-6.
        e, worst, num, den2, per = 0.0, "", 0.0, 0.0, []
        dot, pn2, tnum, tden = 0.0, 0.0, 0.0, 0.0
        for name, g in refg.items():
            gp, go = grads[key][name].double(), g.double()
            diff = gp - go
            num += float(torch.dot(diff.flatten(), diff.flatten()))
            den2 += float(torch.dot(go.flatten(), go.flatten()))
            dot += float(torch.dot(gp.flatten(), go.flatten()))
            pn2 += float(torch.dot(gp.flatten(), gp.flatten()))
            if "outlogits.3." in name:            # the critics' logit layer: behind every activation of the net
                tnum += float((diff * diff).sum())
                tden += float((g.double() * g.double()).sum())
            per.append((float(torch.dot(diff.flatten(), diff.flatten())), name, float(diff.norm() / max(float(go.norm()), 1e-30)), float(go.norm())))
            del gp, go
            # (floor: 5e-3 of the net's largest gradient entry - a one-element tensor whose true value nearly cancels, like the logit
            # layer's bias gradient sum(dz) = 3.5e-4 in the cascade fixture, is measured against the net's scale, not its own)
            ei = diff.abs().max().item() / max(g.abs().max().item(), 5e-3 * scale)
            if ei > e:
                e, worst = ei, "%s(max|ref|=%.2e,nbad=%d/%d)" % (name, g.abs().max().item(),
                                                               int((diff.abs() > 1e-3 * g.abs().max()).sum()), g.numel())
        rep["grad_" + key] = e
        rep["gradl2_" + key] = (num / max(den2, 1e-30)) ** 0.5
        rep["gradcos_" + key] = dot / max((pn2 * den2) ** 0.5, 1e-30)
        if tden:
            rep["gradtail_" + key] = (tnum / tden) ** 0.5
        rep["worst_" + key] = worst
        per.sort(reverse=True)
        rep["worst_top_" + key] = "; ".join("%s share=%.2f rel=%.1e |g|=%.1e" % (n, sq / max(num, 1e-300), rl, gn) for sq, n, rl, gn in per[:5])
    return rep


def assert_step(rep, dtype, scale=1.0, g_elem=1.0):
    """g_elem: extra factor on the PER-ELEMENT bound of the generator's gradient only (bf16 at the fixtures' 2-64 channel widths:
    one BatchNorm output within round-off of zero flips one ReLU mask and moves single elements of one channel's gradient by
    ~its tensor's max while the whole-vector L2 error stays put; measured worst element / tensor max, r03 and r04 builds:
    plain 0.43-0.54 (round 6's fixture: 0.81), cascade 0.54-1.03, order critic 1.51-2.24; the critics' nets 0.14-0.45 in all three. The order-critic figure
    is chaotic in the literal sense: three builds of the BatchNorm kernels that differ by an ulp in ONE product - the activation
    derivative through a switch, through a select, and with the pre-activation recomputed exactly as the forward computes it -
    give 1.52 / 2.24 / 2.19 there while that fixture's whole-vector L2 goes 0.638 / 0.628 / 0.566)."""
    ltol, l2tol, gtol = (t * scale for t in STEP_TOL[dtype])
    assert rep["loss_rel"] < ltol, rep
    assert rep["acc_abs"] < (1e-6 if dtype == "fp32" else 0.35), rep
    for k, v in rep.items():
        if k.startswith("gradl2_"):
            assert v < l2tol, rep
        elif k.startswith(("gradcos_", "gradtail_")):
            continue
        elif k.startswith("grad_"):
            assert v < gtol * (g_elem if k == "grad_G" else 1.0), rep


def state_error(product_net, oracle_net, lr, steps=1):
    """Post-step state, product vs oracle, full tensors: parameters against the Adam bound (a step moves an entry by
    at most ~lr; where the true gradient is 0 its sign is round-off), buffers relative to their own scale.
    Returns (worst parameter deviation / (lr*steps), worst BatchNorm-buffer relative error, names); the worst spectral-norm
    u/v error is left in state_error.last_sn. u/v get their own (looser) bound: the scoring pass of the G step runs
    the power iteration on the critic weights AFTER their Adam step, and entries whose gradient is round-off (the head
    conv's columns for conditioning inputs that are zero in this batch) moved by +-lr on its sign - a few-percent
    change of those columns that v follows."""
    params = {k for k, _ in oracle_net.named_parameters()}
    osd, psd = oracle_net.state_dict(), product_net.state_dict()
    assert set(osd) == set(psd)
    wp, wb, np_, nb, wsn = 0.0, 0.0, "", "", 0.0
    for name, want in osd.items():
        got = psd[name].detach().cpu()
        if name.endswith("num_batches_tracked"):
            assert int(got) == int(want), (name, int(got), int(want))
            continue
        d = (got.double() - want.double()).abs().max().item()
        if name in params:
            if d / (lr * steps) > wp:
                wp, np_ = d / (lr * steps), name
        else:
            drift = 2.2 * lr * (steps - 1) if name.endswith("running_mean") else 0.0     # see golden_util.check_after_state
            e = max(d - drift, 0.0) / (want.abs().max().item() + 1e-12)
            if name.endswith(("weight_u", "weight_v")):
                wsn = max(wsn, e)
            elif e > wb:
                wb, nb = e, name
    state_error.last_sn = max(getattr(state_error, "last_sn", 0.0), wsn)
    return wp, wb, np_, nb


class _HostView:
    """state_dict()/named_parameters() of a product net as CPU tensors (what golden_util.check_after_state reads)."""

    def __init__(self, net):
        self._sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        self._pn = [k for k, _ in net.named_parameters()]

    def state_dict(self):
        return self._sd

    def named_parameters(self):
        return [(k, self._sd[k]) for k in self._pn]


def sync_from_oracle(tr, st):
    """Copy the oracle's whole training state into the product trainer: weights, buffers, Adam moments and step
    counts (host mirror + device scalar). The next product step then starts from exactly the oracle's state."""
    import torch
    pairs = ((tr.nets[0], st.netG, tr.optimizerG, st.optG), (tr.nets[1], st.netD_im, tr.im_optimizerD, st.optD_im),
             (tr.nets[2], st.netD_st, tr.st_optimizerD, st.optD_st), (tr.nets[3], st.netD_se, tr.se_optimizerD, st.optD_se))
    for pnet, onet, popt, oopt in pairs:
        pnet.load_state_dict(onet.state_dict())
        oparams = dict(onet.named_parameters())
        for name, p in pnet.named_parameters():
            ost = oopt.state[oparams[name]]
            pst = popt.state[p]
            if "exp_avg" not in pst:                       # a trainer that has not stepped yet
                pst["exp_avg"], pst["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
            pst["exp_avg"].copy_(ost["exp_avg"])
            pst["exp_avg_sq"].copy_(ost["exp_avg_sq"])
        step = int(next(iter(oopt.state.values()))["step"])
        for gi, grp in enumerate(popt.param_groups):
            grp["step"] = step
            if gi in popt._hypers:                          # (created from grp["step"] at the first step otherwise)
                popt._hypers[gi][0][0] = float(step)
    torch.cuda.synchronize()


def run_step_parity(tag="plain", dtype="fp32", check=True, return_names=False, two_stream=False, deterministic=True,
                    batch_passes=None):
    """Product step vs oracle step on the golden fixture (weights, batch and noise from the real reference run).
    Also compared: the no-grad pass outputs against the REFERENCE's own (fixture nograd/*), accuracies, and the whole
    post-step state (post-Adam parameters, SN u/v, BN running statistics) against the oracle's and the reference's
    summaries (fixture after/*). Returns max relative errors; asserts tolerances when check=True."""
    from oracle.cpcsv_oracle import NoiseTape, train_step
    from cpcsv import runtime
    fx = gu.load("step_%s.npz" % tag)
    oc, st, sds = oracle_state_for(fx)
    stb, imb = gu.batches(fx)
    tape = gu.noise_tape(fx)
    plan = gu.shuffle_plan_of(fx)
    with oracle_threads(fx):
        ref = train_step(st, stb, imb, noise=NoiseTape(tape), shuffle=plan)       # oracle (CPU fp32)
    import miscc.utils as MU
    MU.shuffle_plan_source = (lambda b, t: plan) if plan is not None else None
    keep_batch = MU.BATCH_PASSES
    if batch_passes is not None:       # False: one launch set per reference call (the pre-round-3 path, still the fallback)
        MU.BATCH_PASSES = bool(batch_passes)
    if two_stream:
        MU.BATCH_PASSES = False
    # deterministic=False: the DEFAULT kernel configuration (what bench.py times): weight gradients with pixel splits and
    # float atomics, BatchNorm / spectral-norm / bias sums through contended atomics
    was = runtime.set_deterministic(deterministic)
    try:
        tr = make_trainer(oc, sds, dtype)                          # product (HIP)
        netG = tr.nets[0]
        if two_stream:
            _force_two_stream_nograd(tr)
        set_noise(netG, TapeSource(tape))
        grads = {}
        hooks = _capture_grads(tr, grads)
        seen = {}
        orig_ng = tr._nograd_fakes

        def spy(*a):
            r = orig_ng(*a)
            seen.update(zip(("st_fake", "c_mu", "im_fake", "cim_mu", "se_fake"), r))
            return r
        tr._nograd_fakes = spy
        out = tr.train_step(to_dev(stb), to_dev(imb))
        torch.cuda.synchronize()
        for h in hooks:
            h()
    finally:
        runtime.set_deterministic(was)
        MU.shuffle_plan_source = None
        MU.BATCH_PASSES = keep_batch
    rep = compare_step(out, ref, grads, oc.cascade, seq=oc.use_seq_consistency)
    # ... and the same comparison against the REFERENCE's own record of this step, no oracle in between: every scalar it logged
    # (fixture scalar/*) and every gradient tensor it produced (fixture grad/<net>/*; the order critic's 4.6 M weights are stored as
    # summaries only and are left to the oracle comparison above)
    direct = {k[len("scalar/"):]: float(fx[k]) for k in fx.files if k.startswith("scalar/")}
    for key, gk in NETKEYS:
        direct[gk] = gu.group(fx, "grad/" + key)
    rep_ref = compare_step(out, direct, grads, oc.cascade, seq=oc.use_seq_consistency)
    rep["vs_reference"] = {k: v for k, v in rep_ref.items() if not k.startswith("worst")}
    rep["nograd"] = max(gu.rel_err(seen[k].contiguous(), fx["nograd/" + k]) for k in seen)
    lrs = {"G": oc.g_lr, "D_im": oc.d_lr, "D_st": oc.d_lr, "D_se": oc.d_lr}
    onets = {"G": st.netG, "D_im": st.netD_im, "D_st": st.netD_st, "D_se": st.netD_se}
    rep["param_dev_lr"], rep["buffer_rel"] = 0.0, 0.0
    state_error.last_sn = 0.0
    for pnet, key in zip(tr.nets, ("G", "D_im", "D_st", "D_se")):
        wp, wb, n1, n2 = state_error(pnet, onets[key], lrs[key])
        if wp > rep["param_dev_lr"]:
            rep["param_dev_lr"], rep["worst_param"] = wp, key + "." + n1
        if wb > rep["buffer_rel"]:
            rep["buffer_rel"], rep["worst_buffer"] = wb, key + "." + n2
    rep["sn_uv_rel"] = state_error.last_sn
    if check:
        # clevr: ST=2, BatchNorm1d over two rows; seq: the generator's gradient through the order critic's MSE passes
        # BatchNorm over 3 / 12 values (oracle-vs-reference itself: 1.5e-3) - tests/test_oracle_vs_golden.py
        loose = {"clevr": 20.0, "seq": 4.0}.get(tag, 1.0)
        # bf16 + order critic: the generator's gradient additionally runs through the MSE of two order logits, each behind a
        # 10-conv (2+1)D tower with BatchNorm over 6 stories - measured 0.63 relative L2 at the fixture's 2-64 channel widths
        # (losses 1.3 %, critics' gradients inside the ordinary band): its band is 2.3x wider
        assert_step(rep, dtype, scale=loose if dtype == "fp32" else (2.3 if tag == "seq" else 1.0),
                    g_elem=1.0 if dtype == "fp32" else {"plain": 1.5, "cascade": 2.0, "seq": 1.7}.get(tag, 1.0))
        assert_step(rep_ref, dtype, scale=loose if dtype == "fp32" else (2.3 if tag == "seq" else 1.0),
                    g_elem=1.0 if dtype == "fp32" else {"plain": 1.5, "cascade": 2.0, "seq": 1.7}.get(tag, 1.0))
        assert rep["nograd"] < (2e-4 if dtype == "fp32" else 6e-2), rep
        assert rep["param_dev_lr"] < 2.2, rep                                   # every entry within one Adam step
        assert rep["buffer_rel"] < (3e-3 * loose if dtype == "fp32" else 8e-2), rep
        assert rep["sn_uv_rel"] < (3e-2 if dtype == "fp32" else 0.15), rep
        if dtype == "fp32":                                                     # ... and against the reference's own record
            for pnet, key in zip(tr.nets, ("G", "D_im", "D_st", "D_se")):
                gu.check_after_state(fx, "after/" + key, _HostView(pnet), lrs[key], steps=1, buf_rtol=3e-3 * loose, sn_rtol=3e-2)
    if not return_names:
        rep = {k: v for k, v in rep.items() if not k.startswith("worst")}
    return rep


def oracle_snapshot(st):
    """Everything the oracle's next step starts from (weights, buffers, Adam moments / step counts), detached copies."""
    import copy
    nets = (st.netG, st.netD_im, st.netD_st, st.netD_se)
    opts = (st.optG, st.optD_im, st.optD_st, st.optD_se)
    return [(copy.deepcopy(n.state_dict()), copy.deepcopy(o.state_dict())) for n, o in zip(nets, opts)]


def oracle_step_fp64(oc, snap, stb, imb, tape, **kw):
    """The oracle's step from `snap` (oracle_snapshot) evaluated in DOUBLE precision on the same batch and noise
    (tools/oracle_host_check.py: on one host the fp32 and the fp64 oracle agree to 1e-6 at every step of the steps3 fixture, so the
    host-to-host differences of the free-running oracle are differences of STATE - Adam's first step is lr * sign(g) - not of
    one step's arithmetic)."""
    from oracle.cpcsv_oracle import NoiseTape, make_state, train_step
    torch.set_default_dtype(torch.float64)
    try:
        st64 = make_state(oc)
        dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
        for (nsd, osd), net, opt in zip(snap, (st64.netG, st64.netD_im, st64.netD_st, st64.netD_se),
                                        (st64.optG, st64.optD_im, st64.optD_st, st64.optD_se)):
            net.load_state_dict({k: dbl(v) for k, v in nsd.items()})
            osd = {"state": {i: {k: dbl(v) for k, v in s_.items()} for i, s_ in osd["state"].items()}, "param_groups": osd["param_groups"]}
            opt.load_state_dict(osd)
        d = lambda b: {k: dbl(v) for k, v in b.items()}
        return train_step(st64, d(stb), d(imb), noise=NoiseTape([t.double() for t in tape]), **kw)
    finally:
        torch.set_default_dtype(torch.float32)


# NEAR-KINK ELEMENTS (fp32, lock-step runs). At the fixtures' 2-64 channel widths a step evaluates ~1.3 M pre-activations; the ones
# that THIS host's fp32 oracle leaves only a few of its own round-off errors away from zero (oracle/conditioning.py: |z64| / |z32 -
# z64| against the fp64 evaluation of the same step) can land on either side of their ReLU / LeakyReLU kink in another correct fp32
# evaluation - and one flipped element of an n-element layer is worth ~1/sqrt(n) of its gradient (1.8e-2 behind the story critic's
# 3-sample head BatchNorm). The fixtures' seeds were searched for large safety (their meta/kink_safety), but the oracle's state at
# steps 1 and 2 is the HOST's, so a lock-step run may still meet such an element. It is then resolved, not tolerated: the oracle's
# step is re-evaluated with near-kink elements put on the other side (conditioning.match_kink_sides: only elements closer than 16
# round-offs, at most 12, each flip kept only if it brings the oracle closer to the product) and the product is held to the SAME
# single-step bands against that evaluation. No wider band exists anywhere; a deviation that no assignment of sides to the listed
# elements explains fails. CPCSV_KINK_MATCH=0 switches the re-evaluation off (the bands then apply to the host's own sides).
import os as _os

KINK_MATCH = _os.environ.get("CPCSV_KINK_MATCH", "1") != "0"


def _within(rep, dtype):
    try:
        assert_step(rep, dtype)
        return True
    except AssertionError:
        return False


def resolve_kinks(oc, snap, stb, imb, tape, out, grads, rep):
    """-> (ref, state after the step, rep, kept flips) of the oracle evaluation that agrees with the product's step (out, grads), or
    None if no listed element's side brings the two closer."""
    from oracle import conditioning as COND
    err = lambda ref: sum(v for k, v in compare_step(out, ref, grads, oc.cascade).items() if k.startswith("gradl2_"))
    ref2, st2, kept = COND.match_kink_sides(oc, snap, stb, imb, tape, err)
    if not kept:
        return None
    for name, numel, safety in kept:
        print("near-kink element resolved: %s (%d elements), %.2f fp32 round-offs from zero on this host" % (name, numel, safety))
    return ref2, st2, compare_step(out, ref2, grads, oc.cascade), kept


def run_multistep_parity(tag="plain", dtype="fp32", lockstep=True, check=True):
    """K=3 consecutive steps on the steps3 fixture (fresh batch and noise per step; Adam at t=1,2,3, SN u/v and BN
    running statistics carried over).
    lockstep=True : after every step the oracle's state (weights, buffers, Adam moments) is copied into the product, so
                    step k tests the product's transition function from the SAME state the oracle is in - single-step
                    tolerances hold at every k.
    lockstep=False: both sides run freely; bounds follow the measured oracle-vs-reference divergence (Adam turns
                    round-off on ~0 gradients into +-lr moves, tests/test_oracle_vs_golden.py)."""
    from oracle.cpcsv_oracle import NoiseTape, train_step
    from cpcsv import runtime
    fx3 = gu.load("steps3_%s.npz" % tag)
    fx = gu.load(str(fx3["meta/weights_from"]))
    oc, st, sds = oracle_state_for(fx, gu.cfg_of(fx3))
    was = runtime.set_deterministic(True)
    reps, events = [], []
    try:
        tr = make_trainer(oc, sds, dtype)
        lrs = {"G": oc.g_lr, "D_im": oc.d_lr, "D_st": oc.d_lr, "D_se": oc.d_lr}
        onets = {"G": st.netG, "D_im": st.netD_im, "D_st": st.netD_st, "D_se": st.netD_se}
        for k in range(int(fx3["meta/steps"])):
            pre = "s%d/" % k
            stb, imb = gu.batches(fx3, pre)
            tape = gu.noise_tape(fx3, pre)
            snap = oracle_snapshot(st)
            with oracle_threads(fx3):
                ref = train_step(st, stb, imb, noise=NoiseTape(tape))
            set_noise(tr.nets[0], TapeSource(tape))
            grads = {}
            hooks = _capture_grads(tr, grads)
            out = tr.train_step(to_dev(stb), to_dev(imb))
            torch.cuda.synchronize()
            for h in hooks:
                h()
            rep = compare_step(out, ref, grads, oc.cascade)
            if lockstep and dtype == "fp32" and KINK_MATCH and not _within(rep, dtype):
                with oracle_threads(fx3):
                    got = resolve_kinks(oc, snap, stb, imb, tape, out, grads, rep)
                if got is not None:
                    ref, st, rep, kept = got
                    onets = {"G": st.netG, "D_im": st.netD_im, "D_st": st.netD_st, "D_se": st.netD_se}
                    events += kept
            rep["param_dev_lr"], rep["buffer_rel"] = 0.0, 0.0
            state_error.last_sn = 0.0
            for pnet, key in zip(tr.nets, ("G", "D_im", "D_st", "D_se")):
                wp, wb, n1, n2 = state_error(pnet, onets[key], lrs[key], steps=1 if lockstep else k + 1)
                rep["param_dev_lr"], rep["buffer_rel"] = max(rep["param_dev_lr"], wp), max(rep["buffer_rel"], wb)
            rep["sn_uv_rel"] = state_error.last_sn
            reps.append(rep)
            if check:
                if lockstep:
                    assert_step(rep, dtype)
                    assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < (3e-3 if dtype == "fp32" else 8e-2), (k, rep)
                    assert rep["sn_uv_rel"] < (3e-2 if dtype == "fp32" else 0.15), (k, rep)
                else:
                    # (step 1 measured 1.6e-3 on st_G: the story critic's head BatchNorm sees 3 samples, and its first Adam step
                    # turned round-off-sized gradient entries into +-lr moves)
                    assert rep["loss_rel"] < (2e-4, 3e-3, 1e-2)[k] * (1 if dtype == "fp32" else 100), (k, rep)
                    for key, _ in NETKEYS:
                        assert rep["gradl2_" + key] < (5e-3, 0.1, 0.3)[k] * (1 if dtype == "fp32" else 4), (k, rep)
                    assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < (3e-3, 1e-2, 3e-2)[k] * (1 if dtype == "fp32" else 20), (k, rep)
                    assert rep["sn_uv_rel"] < (3e-2, 6e-2, 0.1)[k] * (1 if dtype == "fp32" else 3), (k, rep)
            if lockstep:
                sync_from_oracle(tr, st)
    finally:
        runtime.set_deterministic(was)
    return reps


def _capture_grads(tr, store):
    """Snapshot .grad of each net right before its optimiser step (after the step zero_grad may clear it)."""
    restore = []
    _opt_of = {"G": tr.optimizerG, "D_im": tr.im_optimizerD, "D_st": tr.st_optimizerD, "D_se": tr.se_optimizerD}
    for key, opt, net in (("G", tr.optimizerG, tr.nets[0]), ("D_im", tr.im_optimizerD, tr.nets[1]),
                          ("D_st", tr.st_optimizerD, tr.nets[2]), ("D_se", tr.se_optimizerD, tr.nets[3])):
        orig = opt.step

        def wrapped(closure=None, _k=key, _n=net, _o=orig, **kw):
            for chunk in (kw.get("pending") or ()):          # data-parallel runs: accumulator chunks still on the wire
                chunk[3]()
            gs = kw.get("gscale", 1.0)                       # ... and SUM-reduced: the mean is folded into the update kernel
            store[_k] = grads_of(_n, _opt_of[_k], kw.get("wire_of") if kw.get("pending") else None)
            if gs != 1.0:
                fused = {name for name, p in _n.named_parameters() if _opt_of[_k].is_fused(p)}
                for name in fused:
                    store[_k][name] *= gs
            return _o(closure, **kw) if closure is not None else _o(**kw)
        opt.step = wrapped
        restore.append(lambda _opt=opt, _orig=orig: setattr(_opt, "step", _orig))
    return restore
