"""The BASELINE configuration at its real sizes (cfg/final.yml widths, ST=12 stories / IM=60 images, 64x64).

The oracle cannot run a whole step of this size in test time, so the full-size checks are (a) single layers at their
real widths against the CPU fp32 reference of that layer and (b) properties that do not depend on the size:
exact linearity of the gather-GEMM in its input, BatchNorm output statistics, agreement of the bf16 step with the
fp32 step (same weights, batch and noise), agreement of the graph-replayed step with the eagerly launched one, and
finiteness of every loss / gradient / running statistic after several steps."""
import os
import types

import pytest
import torch

from tests import op_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", sorted(op_cases.FULL_CASES))
def test_fullsize_layer_matches_torch(name, dtype):
    """Errors are max |a-b| over the tensor / max |b|. Forward and buffers: 2e-4 (fp32) / 2e-2 (bf16). Gradients: with
    millions of pre-activations a handful lie within round-off of zero and get the other ReLU/LeakyReLU slope than in
    the CPU reference (its conv differs in the last ulp); one such flip moves single elements of dx by up to ~1 % of
    the tensor's max while every reduction stays at 1e-6 (measured: only BN+activation layers, only at >= 3 M
    elements) - hence 2e-2 for fp32 gradients here, 0.25 for bf16 as in the small cases."""
    rep = op_cases.run_case(name, dtype)
    ftol, gtol = op_cases.tolerances(dtype)
    gtol = max(gtol, 2e-2)
    for k, v in rep.items():
        assert v < (ftol if (k == "y" or k.startswith("buf_")) else gtol), (name, dtype, rep)


def test_fullsize_gemm_is_exactly_linear_and_bn_normalises():
    """conv(2x) == 2 conv(x) bit for bit (a power-of-two scale commutes with every rounding in the MFMA path), and the
    fused BatchNorm's pre-activation output has zero mean / unit variance per channel - at 60 x 32x32 x 512."""
    from cpcsv import functional as F, modules as M, runtime
    runtime.set_compute_dtype("bf16")
    torch.manual_seed(0)
    conv = M.FusedSequential(M.Conv2d(512, 256, 3, 1, 1, bias=False)).cuda()
    x = torch.randn(60, 512, 32, 32, device="cuda")
    with torch.no_grad():
        a = F.ToPlanarFn.apply(conv(F.ToNhwcFn.apply(x, runtime.tdtype())), 256)
        b = F.ToPlanarFn.apply(conv(F.ToNhwcFn.apply(2 * x, runtime.tdtype())), 256)
        assert torch.equal(2 * a, b)
        blk = M.FusedSequential(M.Conv2d(512, 256, 3, 1, 1, bias=False), M.BatchNorm2d(256)).cuda()
        y = F.ToPlanarFn.apply(blk(F.ToNhwcFn.apply(x, runtime.tdtype())), 256).float()
    m, v = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    assert m.abs().max().item() < 2e-2 and (v - 1).abs().max().item() < 3e-2, (m.abs().max().item(), (v - 1).abs().max().item())


def _trainer(dtype, st=12, im=60):
    import bench
    from cpcsv import runtime
    runtime.set_compute_dtype(dtype)
    bench.pororo_cfg(st, im)
    import trainer as T
    torch.manual_seed(0)
    tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
    tr.setup()
    return tr, bench.synthetic_batches(st, im, 1, "cuda")


def test_store_first_accumulators_are_fully_overwritten():
    """cpcsv.dist.GradBucket.zero() skips the fill of the deferred-update layers whose first weight-gradient launch of a step stores
    (one pixel slice) instead of adding with float atomics - 0.5 of the 0.62 GB of fills per step at these widths. Here every skipped
    accumulator is NaN-filled instead (dist._POISON_SKIPPED): five default-mode steps at the benchmark's widths must leave every
    weight, moment and loss finite - each such launch overwrites its whole accumulator, pads included - most layers must indeed
    be skipped, and the pixel-split ones must not."""
    from cpcsv import dist as cdist, runtime
    assert not runtime.deterministic() and cdist._STORE_FIRST
    keep = cdist._POISON_SKIPPED
    cdist._POISON_SKIPPED = True
    try:
        tr, (stb, imb) = _trainer("bf16")
        for _ in range(5):
            out = tr.train_step(stb, imb)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in out.values() if torch.is_tensor(v)), {k: float(v) for k, v in out.items() if torch.is_tensor(v)}
        flags = []
        for key, bucket in tr._buckets.items():
            for lay in bucket.__dict__.get("fused_layers", ()):
                flags.append((key, lay.name, bool(getattr(lay, "store_first", False))))
        stored = [f for f in flags if f[2]]
        assert len(stored) >= len(flags) // 2 and len(stored) < len(flags), flags      # both kinds exist at these widths
        for n in tr.nets:
            for k, v in n.state_dict().items():
                assert torch.isfinite(v.float()).all(), k
        for opt in (tr.optimizerG, tr.im_optimizerD, tr.st_optimizerD, tr.se_optimizerD):
            for st_ in opt.state.values():
                for k, v in st_.items():
                    if torch.is_tensor(v):
                        assert torch.isfinite(v.float()).all(), k
    finally:
        cdist._POISON_SKIPPED = keep


def _fixed_noise():
    bank = {}

    def draw(shape):
        if shape not in bank:
            g = torch.Generator().manual_seed(1000 + len(bank))
            bank[shape] = torch.randn(shape, generator=g).cuda()
        return bank[shape]
    return draw


_FULLWIDTH = {}


def _fullwidth_oracles(st=3, im=9, cascade=False, fp64=True, **cfg_kw):
    """The oracle at cfg/final.yml WIDTHS (ngf 2048, seg 1024, ndf 124, text 356, T=5), ST=3/IM=9, one step from its
    seeded init: once in fp32 (the reference's arithmetic) and once in fp64 on the same weights, batch and noise.
    The fp64 run is the yardstick: at these widths the step is ill-conditioned (BatchNorm1d over ST rows in the text
    encoders; 32768-feature BatchNorm1d over 15 rows), the fp32 oracle's own generator gradient is only good to ~6 %
    against fp64 (measured here, CPU), so 'product vs fp32 oracle' alone cannot tell a kernel error from round-off."""
    ck = (cascade, st, im, tuple(sorted(cfg_kw.items())))
    if ck in _FULLWIDTH:
        return _FULLWIDTH[ck]
    import copy
    from oracle.cpcsv_oracle import NoiseTape, make_state, pororo_cfg, synthetic_batch, train_step
    oc = pororo_cfg(st_batch=st, im_batch=im, cascade=cascade, **cfg_kw)
    state = make_state(oc, seed=0)
    names = ("G", "D_im", "D_st", "D_se")
    nets = lambda s_: (s_.netG, s_.netD_im, s_.netD_st, s_.netD_se)
    sds = {k: copy.deepcopy(n.state_dict()) for k, n in zip(names, nets(state))}
    stb, imb = synthetic_batch(oc, seed=1)
    torch.manual_seed(5)
    # 32 CPU threads for the oracle: on the MI355X boxes' 256-core hosts MKL-DNN is fastest at 16-32 threads on these layer sizes and
    # 40x slower at 256 (profiles/r05_cpu_threads.txt)
    keep_threads = torch.get_num_threads()
    torch.set_num_threads(min(32, max(1, os.cpu_count() or 1)))
    ref32 = train_step(state, stb, imb, noise=NoiseTape())
    if not fp64:          # (a caller that only needs the fp32 oracle: the fp64 entries alias it)
        torch.set_num_threads(keep_threads)
        _FULLWIDTH[ck] = dict(oc=oc, sds=sds, stb=stb, imb=imb, ref32=ref32, ref64=ref32, state32=state)
        return _FULLWIDTH[ck]
    torch.set_default_dtype(torch.float64)
    try:
        st64 = make_state(oc, seed=0)
        for k, n in zip(names, nets(st64)):
            n.load_state_dict(sds[k])
        d = lambda b: {k: v.double() for k, v in b.items()}
        ref64 = train_step(st64, d(stb), d(imb), noise=NoiseTape([t.double() for t in ref32["noise_tape"]]))
    finally:
        torch.set_default_dtype(torch.float32)
        torch.set_num_threads(keep_threads)
    _FULLWIDTH[ck] = dict(oc=oc, sds=sds, stb=stb, imb=imb, ref32=ref32, ref64=ref64, state32=state)
    return _FULLWIDTH[ck]


def _grad_l2(got, want):
    num = den = 0.0
    for n, g in want.items():
        d = got[n].double().cpu() - g.double()
        num += float((d * d).sum())
        den += float((g.double() ** 2).sum())
    return (num / max(den, 1e-300)) ** 0.5


def fullwidth_vs_oracle(dtype, cascade=False, st=3, im=9, fp64=True, **cfg_kw):
    """One product step at the benchmark's widths against the fp64 oracle; also returns the fp32 ORACLE's error against
    fp64 (the accuracy the reference's own arithmetic has on this problem)."""
    from cpcsv import runtime
    from tests import parity_util as pu
    o = _fullwidth_oracles(st=st, im=im, cascade=cascade, fp64=fp64, **cfg_kw)
    oc, ref32, ref64 = o["oc"], o["ref32"], o["ref64"]
    was = runtime.set_deterministic(True)
    try:
        tr = pu.make_trainer(oc, o["sds"], dtype)
        pu.set_noise(tr.nets[0], pu.TapeSource(ref32["noise_tape"]))
        grads = {}
        hooks = pu._capture_grads(tr, grads)
        out = tr.train_step(pu.to_dev(o["stb"]), pu.to_dev(o["imb"]))
        torch.cuda.synchronize()
        for h in hooks:
            h()
    finally:
        runtime.set_deterministic(was)
    rep = {}
    worst = 0.0
    lnames = dict(pu.LOSS_NAMES)
    if cascade:
        lnames.update(pu.CASCADE_NAMES)
    for rk, pk in lnames.items():
        worst = max(worst, abs(float(out[pk]) - float(ref64[rk])) / (abs(float(ref64[rk])) + 1e-8))
    rep["loss_rel"] = worst
    rep["oracle32_loss_rel"] = max(abs(float(ref32[rk]) - float(ref64[rk])) / (abs(float(ref64[rk])) + 1e-8) for rk in lnames)
    for key, gk in pu.NETKEYS:
        rep["gradl2_" + key] = _grad_l2(grads[key], ref64[gk])
        rep["oracle32_gradl2_" + key] = _grad_l2(ref32[gk], ref64[gk])
        # direction and length against fp64: cos of the whole gradient vector, and |got| / |want|
        dot = sum(float((grads[key][n].double().cpu() * g.double()).sum()) for n, g in ref64[gk].items())
        n1 = sum(float((grads[key][n].double() ** 2).sum()) for n in ref64[gk]) ** 0.5
        n2 = sum(float((g.double() ** 2).sum()) for g in ref64[gk].values()) ** 0.5
        rep["cos_" + key], rep["len_" + key] = dot / max(n1 * n2, 1e-300), n1 / max(n2, 1e-300)
    full = pu.compare_step(out, ref32, grads, cascade)
    rep.update({k: v for k, v in full.items() if k.startswith("worst_top")})
    lrs = {"G": oc.g_lr, "D_im": oc.d_lr, "D_st": oc.d_lr, "D_se": oc.d_lr}
    st32 = o["state32"]
    onets = {"G": st32.netG, "D_im": st32.netD_im, "D_st": st32.netD_st, "D_se": st32.netD_se}
    rep["param_dev_lr"], rep["buffer_rel"] = 0.0, 0.0
    pu.state_error.last_sn = 0.0
    for pnet, key in zip(tr.nets, ("G", "D_im", "D_st", "D_se")):
        wp, wb, _, _ = pu.state_error(pnet, onets[key], lrs[key])
        rep["param_dev_lr"], rep["buffer_rel"] = max(rep["param_dev_lr"], wp), max(rep["buffer_rel"], wb)
    rep["sn_uv_rel"] = pu.state_error.last_sn
    record = {(False, 3, 9, 5): "fullwidth_plain", (True, 3, 9, 5): "fullwidth_cascade", (False, 12, 60, 5): "fullwidth_bench",
              (False, 2, 8, 4): "fullwidth_clevr"}.get((cascade, st, im, oc.video_len))
    if record is not None:
        rep.update(_against_reference_record(record, o, out, grads, cascade))
    del tr
    torch.cuda.empty_cache()
    return rep


def _against_reference_record(name, o, out, grads, cascade):
    """The same step as the REFERENCE itself recorded it (tests/golden/<name>.npz, oracle/gen_golden.py --fullwidth: the imported
    reference at these widths; weights by seed - its init and the oracle's are the same tensors - batch (fullwidth_bench: its
    checksums), noise, every scalar, 11-number summaries of every gradient): the fp32 oracle and the product against that record.
    Summaries compare the vector of per-tensor |g| sums and the vector of every tensor's first eight entries, each in relative L2."""
    import numpy as np
    from tests import golden_util as gu, parity_util as pu
    fx = gu.load(name + ".npz")
    if any(k.startswith("batch/") for k in fx.files):
        stb, imb = gu.batches(fx)
        assert all(torch.equal(stb[k], o["stb"][k]) for k in stb) and all(torch.equal(imb[k], o["imb"][k]) for k in imb)
    else:
        for tag, batch in (("st", o["stb"]), ("im", o["imb"])):
            for k, v in batch.items():
                assert np.array_equal(gu.summarise(v), fx["batchsum/%s/%s" % (tag, k)]), (tag, k)
    tape = gu.noise_tape(fx)
    assert len(tape) == len(o["ref32"]["noise_tape"]) and all(torch.equal(a, b) for a, b in zip(tape, o["ref32"]["noise_tape"]))
    wsum = sum(float(v.double().abs().sum()) for sd in o["sds"].values() for v in sd.values())
    assert abs(wsum - float(fx["meta/weights_sum"])) <= 1e-9 * wsum, (wsum, float(fx["meta/weights_sum"]))
    rep = {}
    names = dict(pu.LOSS_NAMES)
    if cascade:
        names.update(pu.CASCADE_NAMES)
    scal = {k[len("scalar/"):]: float(fx[k]) for k in fx.files if k.startswith("scalar/")}
    rel = lambda got, want: abs(float(got) - want) / (abs(want) + 1e-8)
    rep["refrec_loss_oracle32"] = max(rel(o["ref32"][rk], scal[rk]) for rk in names)
    rep["refrec_loss_product"] = max(rel(out[pk], scal[rk]) for rk, pk in names.items())
    for key, gk in pu.NETKEYS:
        ea, eh = gu.grad_summary_error(fx, "gradsum/" + key, o["ref32"][gk])
        rep["refrec_grad_oracle32_" + key] = max(ea, eh)
        ea, eh = gu.grad_summary_error(fx, "gradsum/" + key, grads[key])
        rep["refrec_grad_product_" + key] = max(ea, eh)
    return rep


@pytest.mark.parametrize("which", ["plain", "cascade", "bench", "clevr"])
def test_fullwidth_step_matches_the_reference_record(which):
    """cfg/final.yml widths, fp32: the step as the imported REFERENCE recorded it (fixtures fullwidth_plain / fullwidth_cascade at ST=3
    / IM=9, fullwidth_bench at the BENCHMARKED batch ST=12 / IM=60) against (a) the fp32 oracle - which pins the oracle to the
    reference at the benchmark's widths, not only at the tiny fixtures' - and (b) the product, no oracle in between; fullwidth_clevr:
    the CLEVR dimensions of BASELINE config 1 at the full widths. Bounds from the
    accuracy the problem allows (the fp32 oracle itself is 2-6 % from fp64 in the generator's gradient at these widths,
    test_fullwidth_step_matches_oracle): losses 5e-4; critics' gradient summaries 1e-2; the generator's 0.15 (cascade: 0.25)."""
    kw = {"plain": {}, "cascade": {"cascade": True}, "bench": {"st": 12, "im": 60},
          "clevr": {"st": 2, "im": 8, "video_len": 4, "text_dim": 72, "label_num": 15, "fp64": False}}[which]
    rep = fullwidth_vs_oracle("fp32", **kw)
    print("FULLWIDTH-REFERENCE", which, {k: "%.3g" % v for k, v in rep.items() if k.startswith("refrec_")})
    # clevr = BASELINE config 1's dimensions (T=4, text 72, labels 15, ST=2 / IM=8): BatchNorm1d over TWO story rows has x_hat = +-1 and
    # an inverse standard deviation of 2 / |a - b| - round-off in the story branch is amplified ~10x more than at ST=3 (the tiny
    # step_clevr fixture carries the same factor): losses 5e-3 (measured: product 7.1e-4, oracle 1.6e-4)
    ltol = 5e-3 if which == "clevr" else 5e-4
    for who in ("oracle32", "product"):
        assert rep["refrec_loss_" + who] < ltol, rep
        assert rep["refrec_grad_%s_G" % who] < (0.25 if which == "cascade" else 0.15), rep
        for key in ("D_im", "D_st", "D_se"):
            assert rep["refrec_grad_%s_%s" % (who, key)] < 1e-2, (who, key, rep)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fullwidth_step_matches_oracle(dtype):
    """The benchmarked dtype (bf16) and the parity dtype (fp32) against the oracle at the benchmark's widths: every loss
    term, every network's whole gradient vector (relative L2) and the post-step state.
    fp32: the product must be as accurate against fp64 as the reference's own fp32 arithmetic is. Critics: gradient
    error <= 2x the fp32 oracle's error against fp64 (+5e-3). Generator: <= 0.1 - the fp32 oracle itself is 2.3e-2
    (GPU box host) to 6.2e-2 (8-thread container) away from fp64 depending only on its summation order, the product
    measured 6.0e-2, all of it in the text-encoder layers behind BatchNorm1d over ST=3 rows. Losses within 2e-4.
    bf16 (bf16 MFMA operands, fp32 accumulation, statistics, master weights and Adam; small dense layers in fp32):
    losses within 2 % (measured 0.65 %), critic gradients within 20 % (0.10-0.13), the generator's within 40 % (0.28)
    in relative L2 (profiles/r02_parity.txt has the per-tensor split)."""
    rep = fullwidth_vs_oracle(dtype)
    if dtype == "fp32":
        assert rep["loss_rel"] < 2e-4 + 2 * rep["oracle32_loss_rel"], rep
        assert rep["gradl2_G"] < 0.1, rep
        for key in ("D_im", "D_st", "D_se"):
            assert rep["gradl2_" + key] < 2 * rep["oracle32_gradl2_" + key] + 5e-3, (key, rep)
        assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < 3e-3 and rep["sn_uv_rel"] < 3e-2, rep
    else:
        assert rep["loss_rel"] < 2e-2, rep
        assert rep["gradl2_G"] < 0.4, rep
        for key in ("D_im", "D_st", "D_se"):
            assert rep["gradl2_" + key] < 0.2, (key, rep)
        assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < 5e-2 and rep["sn_uv_rel"] < 0.1, rep


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fullwidth_cascade_step_matches_oracle(dtype):
    """BASELINE config 4's model at the benchmark's widths: cascade_model.StoryGAN (segmentation decoder -> presample +
    four downBlocks -> gates on the image decoder, latent MSE terms and the segmentation auto-encoder, reference
    cascade_model.py:312-320,401-445,528-540; trainer.py:370-384) at cfg/final.yml widths, ST=3/IM=9, one step against the
    fp64 oracle. Same bounds as the plain model (the 128x128 variant of config 4 does not exist in the reference: no oracle)."""
    rep = fullwidth_vs_oracle(dtype, cascade=True)
    if dtype == "fp32":
        assert rep["loss_rel"] < 2e-4 + 2 * rep["oracle32_loss_rel"], rep
        assert rep["gradl2_G"] < 0.1 + 2 * rep["oracle32_gradl2_G"], rep
        for key in ("D_im", "D_st", "D_se"):
            assert rep["gradl2_" + key] < 2 * rep["oracle32_gradl2_" + key] + 5e-3, (key, rep)
        assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < 3e-3 and rep["sn_uv_rel"] < 3e-2, rep
    else:
        # bf16: measured loss 1.9 %, critics 0.10 / 0.23 / 0.10; the generator's gradient has the right length (1.07) but only
        # cos 0.78 against fp64 (relative L2 0.69; 0.53 at ST=12/IM=60) - twice the plain model's error, with or without the
        # streaming thin kernels and with the extra cascade losses switched off: the doubled depth (decoder -> tanh -> encoder
        # -> gates -> decoder) amplifies the bf16 rounding of activations through BatchNorm's mean-removing backward.
        assert rep["loss_rel"] < 4e-2, rep
        assert rep["gradl2_G"] < 0.85 and rep["cos_G"] > 0.7 and 0.9 < rep["len_G"] < 1.15, rep
        for key in ("D_im", "D_st", "D_se"):
            assert rep["gradl2_" + key] < 0.3 and rep["cos_" + key] > 0.95, (key, rep)
        assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < 6e-2 and rep["sn_uv_rel"] < 0.1, rep
    print("FULLWIDTH-CASCADE", dtype, {k: ("%.3g" % v if isinstance(v, float) else v) for k, v in rep.items() if not k.startswith("worst")})


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_benchmark_batch_step_matches_oracle(dtype):
    """The BENCHMARKED batch - ST=12 stories x 5 frames + IM=60 images per rank, cfg/final.yml widths (BASELINE config 2; reference
    trainer.py:255-256,263,277) - one step from the oracle's seeded init against the oracle on the same weights, batch and noise: every
    loss term, every network's whole gradient vector, the post-step state. The yardstick is the fp64 run; the fp32 oracle (the
    reference's own arithmetic) is priced against it too.
    fp32: the product is as accurate as the reference's arithmetic - per network, relative L2 of the gradient against fp64 at most 3x
    the fp32 oracle's own + 1e-3; losses within 2e-4 (+ 2x the oracle's). Measured (round 5): generator 0.026 (fp32 oracle: 0.032),
    critics 4.9e-4 ... 9.1e-4 (oracle: 7.3e-4 ... 7.7e-4).  bf16 (measured: losses 2.6 %, generator 0.216 / cos 0.977, critics
    0.087-0.105 / cos 0.994-0.996): losses within 3.5 %, generator within 0.27 and cos > 0.97, critics within 0.13 and cos > 0.992."""
    rep = fullwidth_vs_oracle(dtype, st=12, im=60)
    print("BENCH-BATCH", dtype, {k: ("%.3g" % v if isinstance(v, float) else v) for k, v in rep.items() if not k.startswith("worst")})
    if dtype == "fp32":
        assert rep["loss_rel"] < 2e-4 + 2 * rep["oracle32_loss_rel"], rep
        for key in ("G", "D_im", "D_st", "D_se"):
            assert rep["gradl2_" + key] < 3 * rep["oracle32_gradl2_" + key] + 1e-3, (key, rep)
        assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < 3e-3 and rep["sn_uv_rel"] < 3e-2, rep
    else:
        assert rep["loss_rel"] < 3.5e-2, rep
        assert rep["gradl2_G"] < 0.27 and rep["cos_G"] > 0.97, rep
        for key in ("D_im", "D_st", "D_se"):
            assert rep["gradl2_" + key] < 0.13 and rep["cos_" + key] > 0.992, (key, rep)
        assert rep["param_dev_lr"] < 2.2 and rep["buffer_rel"] < 5e-2 and rep["sn_uv_rel"] < 0.1, rep
        # ... and against the REFERENCE's own record of this step (fixture fullwidth_bench.npz; summaries): losses 3.5 %, gradient
        # summaries 0.25 (measured: 3.1 %; 0.12 / 0.16 / 0.06 / 0.075)
        assert rep["refrec_loss_product"] < 3.5e-2, rep
        for key in ("G", "D_im", "D_st", "D_se"):
            assert rep["refrec_grad_product_" + key] < 0.25, (key, rep)


def test_trained_state_bf16_gradients_match_fp64_oracle():
    """The bf16 claim on a TRAINED state (the random-init comparisons above are this model's worst case: random BatchNorm + LeakyReLU
    critics give the fakes a rough input-gradient field): 150 product steps in fp32 at cfg/final.yml widths (ST=3 / IM=15), then ONE
    step from that snapshot by the fp64 oracle and by the product in bf16 - same weights, batch, noise (tools/bf16_trained_state.py).
    Measured over five runs: generator gradient 0.057-0.187 relative L2 / cos 0.983-0.9985 against fp64 (the figure depends on the
    state the 150 steps reach, and float atomics make that differ from run to run; the steps of this test run in the
    reproducible-reduction mode: 0.204 / 0.980, bit-identical between runs), critics 0.035-0.051 / 0.999, losses 0.9-1.8 %; at the
    bench batch after 300 steps
    (profiles/r04_bf16_trained_state.txt): 0.064 / 0.998 plain, 0.33 / 0.944 cascade, critics 0.033-0.039."""
    import os
    import sys
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bf16_trained_state as T
    res = T.evaluate(types.SimpleNamespace(st=3, steps=150, cascade=False), arms=("fp32", "bf16"), deterministic=True, frozen=True)
    loss_rel, rows = res["bf16"]
    assert loss_rel < 3e-2, res
    l2, cos, length = rows["G"]
    assert l2 < 0.25 and cos > 0.965 and 0.9 < length < 1.1, rows
    for key in ("D_im", "D_st", "D_se"):
        assert rows[key][0] < 0.08 and rows[key][1] > 0.996, (key, rows)
    # The fp32 arm on the same trained state: as accurate against fp64 as the reference's own fp32 arithmetic (the fp32 oracle).
    # (a) the whole step: the critics at most 3x the oracle's error + 2e-3 (one flipped LeakyReLU mask of a 3-15-sample BatchNorm
    #     is worth 1e-3 of a critic's gradient: the oracle shows such events too - D_st 9.9e-4 in both, round 5); the generator's
    #     gradient is taken through the critics as the step's own Adam update leaves them, so every such flip moves ALL of it: measured
    #     0.0036 and 0.037 on two trained states (oracle: 0.0057 / 0.0070, 0.0016 ... 0.0073 at the bench batch) - bounded at 3x + 5e-2;
    # (b) the same step with the critics' learning rate 0 (tools/bf16_trained_state.evaluate(frozen=True)): the arithmetic of the
    #     generator's forward / backward and of the scoring passes alone - every network at most 3x the oracle's error + 2e-3.
    loss32, rows32 = res["fp32"]
    oloss, orows = res["oracle32"]
    assert loss32 < 2e-3 + 2 * oloss, res
    for key in ("D_im", "D_st", "D_se"):
        assert rows32[key][0] < 3 * orows[key][0] + 2e-3, (key, rows32, orows)
    assert rows32["G"][0] < 3 * orows["G"][0] + 5e-2, (rows32, orows)
    _, frows = res["fp32_frozen"]
    _, forows = res["oracle32_frozen"]
    for key in ("G", "D_im", "D_st", "D_se"):
        assert frows[key][0] < 3 * forows[key][0] + 2e-3, (key, frows, forows)


def test_fullsize_bf16_step_tracks_fp32_step():
    """Same initial weights (seed), batch and noise: every loss of the first bf16 step within 8 % of the fp32 step's
    (bf16 operands, fp32 accumulation through ~40 layers; the widest gap measured is 5 % on the story critic's BCE of
    the fakes, the other terms are within 0.5 %). Later steps are not compared: at these widths and learning rates the
    losses move by factors per step and two runs of the SAME mode drift apart by 10 % within four steps."""
    from tests import parity_util as pu
    hist = {}
    for dtype in ("fp32", "bf16"):
        tr, (stb, imb) = _trainer(dtype)
        pu.set_noise(tr.nets[0], _fixed_noise())
        hist[dtype] = {k: float(v) for k, v in tr.train_step(stb, imb).items() if "Acc" not in k}
        del tr
        torch.cuda.empty_cache()
    for k, a in hist["fp32"].items():
        assert hist["bf16"][k] == pytest.approx(a, rel=8e-2, abs=5e-3), (k, a, hist["bf16"][k])


def test_fullsize_batched_passes_match_separate_launches():
    """At the benchmark size every pair of reference calls that shares weights runs as ONE set of launches (row groups): the
    story half and the image half of a generator pass (both 60 frames; model.StoryGAN.sample_both), a critic's real and
    fake batches, its three head calls. One step from the same weights, batch and noise, deterministic mode, fp32 (exact
    f32 MFMA), batched (default) against one launch set per call (the round-2 path): every loss within 5e-4, the critics'
    gradients (well conditioned) within 2e-3 in relative L2, the generator's accumulators within 0.1 - the level at which
    two fp32 evaluations of this step differ at these widths (only the summation order of the BatchNorm partials differs
    between the two modes; the fp32 oracle itself is 2-6e-2 off its fp64 run)."""
    import gc
    import miscc.utils as MU
    from cpcsv import runtime
    from tests import parity_util as pu
    keep_b = MU.BATCH_PASSES
    was = runtime.set_deterministic(True)
    snaps = {}
    try:
        for mode in ("batched", "separate"):
            MU.BATCH_PASSES = mode == "batched"
            tr, (stb, imb) = _trainer("fp32")
            pu.set_noise(tr.nets[0], _fixed_noise())
            grads = {}
            for key, opt in tr._opt_of.items():
                orig = opt.step

                def grab(closure=None, _o=orig, _k=key, _t=tr):
                    b = _t._buckets[_k]
                    grads[_k] = [t.detach().clone() for t in [b.flat] + list(b.extra)]
                    return _o()
                opt.step = grab
            out = tr.train_step(stb, imb)
            torch.cuda.synchronize()
            grouped = sum(1 for l, w, _ in tr.optimizerG._layers if any(isinstance(k, tuple) and k[0] == "fwd" and k[-1] is not None for k in l.descs))
            assert (grouped > 0) == (mode == "batched"), (mode, grouped)
            snaps[mode] = ({k: float(v) for k, v in out.items() if "Acc" not in k}, grads)
            del tr, grads, out
            gc.collect()
            torch.cuda.empty_cache()
    finally:
        MU.BATCH_PASSES = keep_b
        runtime.set_deterministic(was)
    (la, ga), (lb, gb) = snaps["batched"], snaps["separate"]
    for k in la:
        assert la[k] == pytest.approx(lb[k], rel=5e-4, abs=1e-5), (k, la[k], lb[k])
    for key in ga:
        for a, b in zip(ga[key], gb[key]):
            assert torch.isfinite(a).all() and a.abs().max().item() > 0
            rel = ((a.double() - b.double()).norm() / b.double().norm()).item()
            assert rel < (0.1 if key == "G" else 2e-3), (key, rel)


def test_fullsize_graph_replay_matches_eager_and_stays_finite(monkeypatch):
    """Three steps at the benchmark size with the captured pieces on (one eager warm-up step, then capture + replays)
    and off, same seeds and live RNG: losses within 2 % for the first two steps and 8 % for the third (two eager runs
    differ by ~0.3 % after two steps and several percent after three: fp32 atomics, amplified by the GAN dynamics); all
    gradients, weights and BatchNorm running statistics finite afterwards."""
    monkeypatch.setenv("CPCSV_GRAPH_WARMUP", "1")
    from cpcsv import runtime
    was = runtime.set_deterministic(True)      # fixed-order reductions in both arms: without them the third step's 8 % bound
    try:                                       # is occasionally exceeded by atomic-order noise alone (seen once in four runs)
        res = _graph_vs_eager_runs(monkeypatch)
    finally:
        runtime.set_deterministic(was)
    for i, (a, b) in enumerate(zip(res["0"], res["1"])):
        for k in a:
            assert b[k] == pytest.approx(a[k], rel=2e-2 if i < 2 else 8e-2, abs=3e-3 if i < 2 else 2e-2), (i, k, a[k], b[k])


def test_config5_batch_shape_st32_graph_replay_matches_eager(monkeypatch):
    """BASELINE config 5's per-rank batch (ST=32 stories / IM=160 images: 320 frames per generator pass, 2.7x config 2) in the
    benchmarked dtype. fp8 itself is not built (the reference has no 1x1 convs, its dense GEMMs are 0.6 % of the
    step); what this pins is that nothing in the launch planning (tile choice, split-K plans, row groups, pixel splits, slab
    workspaces, BatchNorm partial counts) is tied to the 12/60 shape: three steps with the captured pieces on and off agree
    like they do at ST=12, and every gradient / weight / running statistic stays finite."""
    monkeypatch.setenv("CPCSV_GRAPH_WARMUP", "1")
    from cpcsv import runtime
    was = runtime.set_deterministic(True)
    try:
        res = _graph_vs_eager_runs(monkeypatch, st=32)
    finally:
        runtime.set_deterministic(was)
    for i, (a, b) in enumerate(zip(res["0"], res["1"])):
        for k in a:
            assert b[k] == pytest.approx(a[k], rel=2e-2 if i < 2 else 8e-2, abs=3e-3 if i < 2 else 2e-2), (i, k, a[k], b[k])


def _graph_vs_eager_runs(monkeypatch, st=12):
    res = {}
    for mode in ("1", "0"):
        for k in ("CPCSV_NOGRAD_GRAPH", "CPCSV_CRITIC_GRAPH", "CPCSV_G_GRAPH", "CPCSV_SCORE_GRAPH"):
            monkeypatch.setenv(k, mode)
        tr, (stb, imb) = _trainer("bf16", st, 5 * st)
        torch.manual_seed(7)
        torch.cuda.manual_seed_all(7)
        res[mode] = [{k: float(v) for k, v in tr.train_step(stb, imb).items() if "Acc" not in k} for _ in range(3)]
        if mode == "1":
            assert tr._ng.captured and tr._gg.captured and all(g.captured for g in tr._cg.values())
            for net in tr.nets:
                for t in list(net.parameters()) + list(net.buffers()):
                    assert torch.isfinite(t).all()
            for b in tr._buckets.values():
                assert all(torch.isfinite(t).all() for t in [b.flat] + b.extra)
        del tr
        torch.cuda.empty_cache()
    return res
