"""GPU parity of single fused layers: HIP LayerFn (through the C ABI) vs plain PyTorch fp32 on CPU."""
import pytest
import torch

from tests import op_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", sorted(op_cases.CASES))
def test_fused_layer_matches_torch(name, dtype):
    rep = op_cases.run_case(name, dtype)
    ftol, gtol = op_cases.tolerances(dtype)
    for k, v in rep.items():
        tol = gtol if k.startswith("d") else ftol
        assert v < tol, (name, dtype, k, v, rep)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", sorted(op_cases.GROUP_CASES))
def test_row_groups_equal_separate_calls(name, dtype):
    """Several passes of one layer in ONE set of launches (cpcsv.runtime.row_groups: real | fake batch of a critic, story |
    image half of a generator pass, real | wrong | fake of a head) against torch running one CALL per pass: outputs, input
    and parameter gradients (accumulated over the passes), BatchNorm running statistics after the sequential updates and
    spectral-norm u / v after one iteration per pass - same tolerances as the single-call cases."""
    rep = op_cases.run_group_case(name, dtype)
    ftol, gtol = op_cases.tolerances(dtype)
    for k, v in rep.items():
        tol = gtol if k.startswith("d") else ftol
        assert v < tol, (name, dtype, k, v, rep)


@pytest.mark.parametrize("name", ["thin_img", "thin_img_ragged_rows", "thin_seg", "thin_enc0", "thin_enc0_seg", "thin_enc0_sn", "thin_enc0_sn_wide"])
def test_thin_kernels_agree_with_the_gather_gemm(name):
    """bf16: the same layer through the streaming kernels of csrc/thin.hip and through the general gather-GEMM. Both
    multiply bf16 operands on the matrix cores with fp32 accumulation, so forward, data gradient and weight gradient
    agree to summation order (bf16 output rounding: 1 ulp = 8e-3 relative, measured against each tensor's max) - far
    tighter than either agrees with the fp32 reference. Also asserts the thin path really was taken."""
    from cpcsv import functional as F
    keep = F._THIN
    try:
        F._THIN = True
        _, a, kinds = op_cases.run_case(name, "bf16", raw=True)
        assert kinds and all(k in (1, 2) for k in kinds), kinds
        F._THIN = False
        _, b, kinds0 = op_cases.run_case(name, "bf16", raw=True)
        assert not kinds0
    finally:
        F._THIN = keep
    for k in a:
        err = (a[k] - b[k]).abs().max().item() / (b[k].abs().max().item() + 1e-12)
        assert err < 1e-2, (name, k, err)


@pytest.mark.parametrize("name", ["full_up2", "full_up4", "full_d_enc2", "full_d_enc4", "full_down1", "full_down4", "full_head", "full_fc"])
def test_wgrad_linear_staging_is_bit_identical(name):
    """bf16 LDS-DMA weight-gradient kernel: linear running-pointer staging (default wherever the map is a power of two wide)
    against its general per-piece gather arithmetic, deterministic mode (one block per tile walks all pixels in order): the
    same MFMAs in the same order, so every parameter gradient must agree bit for bit."""
    from cpcsv import _lib, runtime
    lib = _lib.load()
    was = runtime.set_deterministic(True)
    try:
        keep = lib.cpcsv_set_wgrad_linear(1)
        _, a, _ = op_cases.run_case(name, "bf16", raw=True)
        lib.cpcsv_set_wgrad_linear(0)
        _, b, _ = op_cases.run_case(name, "bf16", raw=True)
    finally:
        lib.cpcsv_set_wgrad_linear(keep)
        runtime.set_deterministic(was)
    for k in a:
        if k.startswith("d_"):
            assert torch.equal(a[k], b[k]), (name, k, (a[k] - b[k]).abs().max().item())


def test_small_ops_match_torch():
    """GRU cell, dynamic filter, reparam, losses, gate, mean_t vs torch (fp32)."""
    import torch.nn as nn
    import torch.nn.functional as TF
    from cpcsv import functional as F
    from cpcsv import modules as M
    from cpcsv import runtime
    from tests import golden_util as gu
    runtime.set_compute_dtype("fp32")
    dev = "cuda"
    torch.manual_seed(3)
    rel = lambda a, b: ((a.detach().float().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12)).item()
    # GRU
    tg = nn.GRUCell(45, 23)
    pg = M.GRUCell(45, 23)
    pg.load_state_dict(tg.state_dict()); pg.to(dev)
    x, h = torch.randn(6, 45), torch.randn(6, 23)
    xt, ht = x.clone().requires_grad_(), h.clone().requires_grad_()
    yt = tg(xt, ht); dy = torch.randn_like(yt); yt.backward(dy)
    xp, hp = x.to(dev).requires_grad_(), h.to(dev).requires_grad_()
    yp = pg(xp, hp); yp.backward(dy.to(dev))
    assert rel(yp, yt) < 1e-5 and rel(xp.grad, xt.grad) < 1e-4 and rel(hp.grad, ht.grad) < 1e-4
    for k, p_ in pg.named_parameters():
        assert rel(p_.grad, dict(tg.named_parameters())[k].grad) < 1e-4, k
    # dynamic filter vs the REAL reference's golden vectors
    fx = gu.load("ops.npz")
    sig = torch.from_numpy(fx["dfl/sig"]).to(dev).requires_grad_()
    taps = torch.from_numpy(fx["dfl/taps"]).to(dev).requires_grad_()
    out = F.DynFilter1dFn.apply(sig, taps.reshape(5, 3, 21), 10)
    out.backward(torch.from_numpy(fx["dfl/up"]).to(dev))
    assert rel(out, torch.from_numpy(fx["dfl/out"])) < 1e-5
    assert rel(sig.grad, torch.from_numpy(fx["dfl/dsig"])) < 1e-5
    assert rel(taps.grad, torch.from_numpy(fx["dfl/dtaps"])) < 1e-5
    # KL vs reference golden
    mu = torch.from_numpy(fx["kl/mu"]).to(dev).requires_grad_(); lv = torch.from_numpy(fx["kl/logvar"]).to(dev).requires_grad_()
    kl = F.KlFn.apply(mu, lv)
    assert abs(kl.item() - float(fx["kl/out"])) < 1e-6 * max(1, abs(float(fx["kl/out"])))
    (kl * 3.0).backward()
    mt = torch.from_numpy(fx["kl/mu"]).requires_grad_(); lt = torch.from_numpy(fx["kl/logvar"]).requires_grad_()
    (-0.5 * torch.mean(1 + lt - mt.pow(2) - lt.exp()) * 3.0).backward()
    assert rel(mu.grad, mt.grad) < 1e-5 and rel(lv.grad, lt.grad) < 1e-5
    # BCE (incl. saturated probabilities -> the -100 clamp) and multilabel soft margin
    p = torch.tensor([0.0, 1.0, 0.3, 0.999999, 1e-8, 0.5]); t = torch.tensor([0.0, 1.0, 1.0, 0.0, 1.0, 0.0])
    pt = p.clone().requires_grad_(); lt_ = TF.binary_cross_entropy(pt, t); lt_.backward()
    pp = p.to(dev).requires_grad_(); lp = F.BceFn.apply(pp, t.to(dev)); lp.backward()
    assert abs(lp.item() - lt_.item()) < 1e-4 * abs(lt_.item()) and rel(pp.grad, pt.grad) < 1e-4
    x = torch.randn(7, 9) * 3; y = (torch.rand(7, 9) < 0.4).float()
    xt = x.clone().requires_grad_(); l1 = TF.multilabel_soft_margin_loss(xt, y); l1.backward()
    xp = x.to(dev).requires_grad_(); l2, acc2 = F.MlsmFn.apply(xp, y.to(dev), 9); l2.backward()
    assert abs(l1.item() - l2.item()) < 1e-5 and rel(xp.grad, xt.grad) < 1e-5
    # ... and get_multi_acc of the same logits out of the same launch (reference miscc/utils.py:313-321)
    want_acc = float(((y == 1) & (torch.sigmoid(x) >= 0.5)).sum()) / float(y.sum())
    assert abs(float(acc2) - want_acc) < 1e-6 and not acc2.requires_grad
    # weighted sum of device scalars, one launch each way (the generator's total loss)
    sc = [torch.tensor(v, device=dev, requires_grad=True) for v in (0.5, -2.0, 3.25)]
    tot = F.LinCombFn.apply([1.0, 5.0, 0.25], *sc)
    tot.backward()
    assert abs(float(tot) - (0.5 - 10.0 + 0.8125)) < 1e-6 and [float(t_.grad) for t_ in sc] == [1.0, 5.0, 0.25]
    # several device-to-device copies in one launch (graph input staging), incl. an odd byte count
    srcs = [torch.randn(7, 13, device=dev), torch.arange(5, device=dev, dtype=torch.int64), torch.randn(3, device=dev).to(torch.bfloat16)] * 4
    dsts = [torch.zeros_like(t_) for t_ in srcs]
    from cpcsv import kernels as K
    K.copy_many(list(zip(dsts, srcs)))
    assert all(torch.equal(a_, b_) for a_, b_ in zip(dsts, srcs))
    # gate, reparam, mean over T, MSE
    a, b = torch.randn(2, 4, 4, 8), torch.randn(2, 4, 4, 8)
    at, bt = a.clone().requires_grad_(), b.clone().requires_grad_()
    g = torch.randn(2, 4, 4, 8); (at * bt + bt).backward(g)
    ap, bp = a.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    F.GateFn.apply(ap, bp).backward(g.to(dev))
    assert rel(ap.grad, at.grad) < 1e-6 and rel(bp.grad, bt.grad) < 1e-6
    xs = torch.randn(6, 2, 2, 8); xp = xs.to(dev).requires_grad_()
    m = F.MeanTFn.apply(xp, 3); m.backward(torch.ones_like(m))
    assert rel(m, xs.view(2, 3, 2, 2, 8).mean(1)) < 1e-6 and abs(xp.grad.mean().item() - 1 / 3) < 1e-6
    a, b = torch.randn(3, 5, 5, 8), torch.randn(3, 5, 5, 8)
    ap = a.to(dev).requires_grad_(); l = F.MseFn.apply(ap, b.to(dev)); l.backward()
    at = a.clone().requires_grad_(); lt2 = TF.mse_loss(at, b); lt2.backward()
    assert abs(l.item() - lt2.item()) < 1e-5 * lt2.item() and rel(ap.grad, at.grad) < 1e-5


def test_fused_adam_matches_torch():
    from cpcsv.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(5000,), (33, 7), (1,), (4097,)]
    ps = [torch.randn(s) for s in shapes]
    a = [p.clone().requires_grad_() for p in ps]
    b = [p.clone().cuda().requires_grad_() for p in ps]
    oa = torch.optim.Adam(a, lr=4e-4, betas=(0.5, 0.999))
    ob = FusedAdam(b, lr=4e-4, betas=(0.5, 0.999))
    for it in range(3):
        for x, y in zip(a, b):
            g = torch.randn_like(x)
            x.grad = g.clone(); y.grad = g.cuda()
        oa.step(); ob.step()
    for x, y in zip(a, b):
        assert (x - y.cpu()).abs().max().item() < 1e-6


@pytest.mark.parametrize("c,hw,frames", [(64, 16, 3), (70, 16, 5), (1024, 16, 4), (130, 64, 2), (96, 1, 7)])
@pytest.mark.parametrize("sd,dd", [("bf16", "bf16"), ("bf16", "f32"), ("f32", "f32")])
def test_nhwc_to_planar_small_maps(c, hw, frames, sd, dd):
    """Many channels on a small map take the LDS-tiled copy (csrc/elementwise.hip nhwc_to_planar_tiled_kernel: the gradient
    entering StoryGAN.fc, reference model.py:250): it must equal the permute it replaces, pad channels ignored."""
    from cpcsv import kernels as K
    td = {"bf16": torch.bfloat16, "f32": torch.float32}
    cs = (c + 7) // 8 * 8
    torch.manual_seed(c + hw)
    src = torch.randn(frames, hw, cs).to(td[sd]).cuda()
    dst = torch.full((frames, c * hw), float("nan"), dtype=td[dd], device="cuda")
    K.nhwc_to_planar(src, dst, frames, 1, c * hw, 0, hw, c, hw, cs)
    torch.cuda.synchronize()
    want = src[:, :, :c].permute(0, 2, 1).reshape(frames, c * hw).to(td[dd])
    assert torch.equal(dst, want)


@pytest.mark.parametrize("c,hw,frames", [(64, 16, 3), (70, 16, 5), (1024, 16, 4), (130, 64, 2), (96, 1, 7), (3, 4096, 2)])
@pytest.mark.parametrize("sd,dd", [("bf16", "bf16"), ("f32", "bf16"), ("f32", "f32")])
def test_planar_to_nhwc_forms(c, hw, frames, sd, dd):
    """planar (C, HW) frames -> NHWC with zero channel pads: the LDS-tiled copy for small maps with many channels (the generator's
    `fc` output, reference model.py:380), the one-thread-per-pixel copy for 8-channel image pixels, the generic one - all equal the
    permute they replace, pads zero over a poisoned destination."""
    from cpcsv import kernels as K
    td = {"bf16": torch.bfloat16, "f32": torch.float32}
    cs = (c + 7) // 8 * 8
    torch.manual_seed(c + hw)
    src = torch.randn(frames, c * hw).to(td[sd]).cuda()
    dst = torch.full((frames, hw, cs), float("nan"), dtype=td[dd], device="cuda")
    K.planar_to_nhwc(src, dst, frames, 1, c * hw, 0, hw, c, hw, cs)
    torch.cuda.synchronize()
    want = torch.zeros(frames, hw, cs, dtype=td[dd], device="cuda")
    want[:, :, :c] = src.view(frames, c, hw).permute(0, 2, 1).to(td[dd])
    assert torch.equal(dst, want)


@pytest.mark.parametrize("m,n,k", [(1, 1, 8), (12, 1095, 368), (17, 63, 40), (60, 372, 1784), (64, 5, 128), (33, 248, 1096)])
def test_dense_rows_kernels(m, n, k):
    """cpcsv_dense_rows / cpcsv_dense_rows_wgrad (the one-launch fp32 dense layers over <= 64 rows: text / motion encoders, GRU
    recurrences, reference model.py:223-224,252-262) against float64 torch: product + bias + activation, zero pads of the output
    row, BatchNorm partials per block of 16 rows, the scaled data-gradient form, and the weight gradient ADDED into dW."""
    from cpcsv import kernels as K, _lib as L
    torch.manual_seed(m * 1000 + n)
    ldy = (n + 7) // 8 * 8
    x, w, b = torch.randn(m, k), torch.randn(n, k), torch.randn(n)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y = torch.full((m, ldy), float("nan"), device="cuda")
    nb = (m + 15) // 16
    stats = torch.full((nb, 2, ldy), float("nan"), device="cuda")
    K.dense_rows(xd, wd, y, m, n, k, None, bd, L.ACT_RELU, stats, ldy)
    t = x.double() @ w.double().t() + b.double()
    tol = 2e-6 * k ** 0.5 * max(1.0, t.abs().max().item())
    assert (y[:, :n].cpu().double() - t.clamp_min(0)).abs().max().item() < tol
    assert (y[:, n:] == 0).all()
    for rb in range(nb):
        blk = t[rb * 16:(rb + 1) * 16]
        assert (stats[rb, 0, :n].cpu().double() - blk.sum(0)).abs().max().item() < 16 * tol
        assert (stats[rb, 1, :n].cpu().double() - (blk * blk).sum(0)).abs().max().item() < 16 * tol * max(1.0, t.abs().max().item())
    alpha = torch.tensor([0.37], device="cuda")
    K.dense_rows(xd, wd, y, m, n, k, alpha, None, L.ACT_NONE)
    assert (y[:, :n].cpu().double() - 0.37 * (x.double() @ w.double().t())).abs().max().item() < tol
    dz = torch.randn(m, ldy)
    dz[:, n:] = 0
    dW = torch.randn(n, k)
    dWd = dW.cuda()
    db = torch.randn(n)
    dbd = db.cuda()
    K.dense_rows_wgrad(dz.cuda(), xd, dWd, m, n, k, dbd)
    want = dW.double() + dz[:, :n].double().t() @ x.double()
    assert (dWd.cpu().double() - want).abs().max().item() < 2e-6 * m ** 0.5 * max(1.0, want.abs().max().item())
    assert (dbd.cpu().double() - (db.double() + dz[:, :n].double().sum(0))).abs().max().item() < 1e-5


def _patch_case(kind, n, hw, cin, cout, groups, seed):
    """Descriptor + operands of one patch-eligible launch. kind 'sub': the sub-pixel form of nearest-x2 upsample + conv3x3
    (4 parity phases x 4 summed taps over the low-resolution grid, scattered output: every upBlock, reference model.py:26-34);
    'tconv': the data gradient of a 4x4 stride-2 pad-1 conv as 4 parity phases x 4 taps over the dY grid (the critics' towers,
    reference model.py:502-513)."""
    from cpcsv import functional as F, kernels as K, _lib as L
    g = torch.Generator().manual_seed(seed)
    cs, cout_s = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
    x = torch.zeros(n, hw, hw, cs)
    x[..., :cin] = torch.randn(n, hw, hw, cin, generator=g)
    x = x.to(torch.bfloat16).cuda()
    w = torch.zeros(cout, 16 * cs)
    w.view(cout, 16, cs)[:, :, :cin] = torch.randn(cout, 16, cin, generator=g) * 0.1
    w = w.to(torch.bfloat16).cuda()
    if kind == "sub":
        taps, phases = F.SUB_FWD_TAPS, F.SUB_PHASES
    else:
        geom = F.ConvGeom(4, 2, 1)
        taps, phases = [], []
        for tp, mh, mw, _, sc in geom.dgrad_launches(2 * hw, 2 * hw):
            assert (mh, mw) == (hw, hw)
            phases.append((len(taps), len(tp), sc[4], sc[5]))
            taps += tp
    if kind in ("conv4", "subd"):
        # 4x4 stride-2 pad-1 window over a 2hw x 2hw input: the critics' tower convs forward (reference model.py:502-513, taps in
        # the parity-class order ConvGeom.fwd_taps gives them) / the data gradient of the sub-pixel upsample+conv
        x = torch.zeros(n, 2 * hw, 2 * hw, cs)
        x[..., :cin] = torch.randn(n, 2 * hw, 2 * hw, cin, generator=g)
        x = x.to(torch.bfloat16).cuda()
        # (the patch-resident loop wants the 16 taps grouped by the parity of the input pixel they read: four runs of four)
        taps = sorted(F.ConvGeom(4, 2, 1).fwd_taps(), key=lambda t: ((t[0] + 1) & 1, (t[1] + 1) & 1)) if kind == "conv4" else F.SUB_DGRAD_TAPS
        y = torch.full((n, hw, hw, cout_s), float("nan"), dtype=torch.bfloat16, device="cuda")
        d = K.gemm_desc(x, w, y, dtype=L.BF16, M=n * hw * hw, N=cout, Cs=cs, ldb=16 * cs, ldc=cout_s, taps=taps, MH=hw, MW=hw,
                        IH=2 * hw, IW=2 * hw, sy=2, sx=2, act=L.ACT_LRELU)
        if groups:
            K.set_row_groups(d, [0] + [c * hw * hw for c in groups])
        return d, x, w, y
    y = torch.full((n, 2 * hw, 2 * hw, cout_s), float("nan"), dtype=torch.bfloat16, device="cuda")
    d = K.gemm_desc(x, w, y, dtype=L.BF16, M=n * hw * hw, N=cout, Cs=cs, ldb=16 * cs, ldc=cout_s, taps=taps, MH=hw, MW=hw, IH=hw, IW=hw,
                    scatter=(2 * hw, 2 * hw, 2, 2, 0, 0), phases=phases, act=L.ACT_RELU)
    if groups:
        K.set_row_groups(d, [0] + [c * hw * hw for c in groups])
    return d, x, w, y


@pytest.mark.parametrize("kind,n,hw,cin,cout,groups", [
    ("sub", 6, 16, 128, 96, None),           # 16x16 maps: one whole image per 256-row tile; N <= 128
    ("sub", 5, 32, 64, 64, (2, 5)),          # 32x32 maps: 8 image rows per tile + halo rows; two row groups; the 64-column tile
    ("sub", 3, 16, 200, 256, (1, 3)),        # channel tail (200 = 3 x 64 + 8), two column tiles
    ("sub", 2, 64, 72, 40, None),            # 64-wide maps (4 image rows per tile, 7 pieces per wave), channel tail
    ("tconv", 7, 16, 248, 124, (3, 7)),      # critics' enc1 data gradient at its real widths: dY 16x16x248 -> dX 32x32x124
    ("conv4", 6, 16, 124, 248, (2, 6)),      # critics' enc1 forward at its real widths: 32x32x124 -> 16x16x248, stride 2 (parity classes)
    ("subd", 3, 32, 64, 128, None),          # data gradient of an up-block: 4x4 stride-2 gather over dY 64x64 -> 32x32
    ("subd", 4, 16, 72, 40, (1, 4)),         # ... with a channel tail, the 64-column tile and row groups
])
def test_patch_resident_main_loop_is_bit_identical(kind, n, hw, cin, cout, groups):
    """conv_patch_kernel (input patch of a 256-row tile resident in LDS, all taps of a phase served from it) against the
    streaming gather-GEMM walking K in the same order (cpcsv_gemm_desc.korder = 1): identical MFMA sequence on identical
    operands -> identical output bits, including the zero padding at the image borders, the rows past a row group's end and
    the channel pads; BatchNorm partials agree as sums (their row partition follows the tile size). And the streaming kernel in
    the taps-outer K order of rounds 1-5 (korder = 2) differs from both only by fp32 summation order."""
    import ctypes as C
    from cpcsv import kernels as K, _lib as L
    lib = L.load()
    outs, sums = {}, {}
    for mode in ("patch", "stream_ct_outer", "stream"):
        d, x, w, y = _patch_case(kind, n, hw, cin, cout, groups, seed=hw * 1000 + cin)
        d.patch, d.korder = (1, 0) if mode == "patch" else ((-1, 1) if mode == "stream_ct_outer" else (-1, 2))
        mt = lib.cpcsv_gemm_mtile(C.byref(d))
        assert (mt == 256) == (mode == "patch") or mode != "patch"
        counts = groups and [groups[0]] + [groups[i] - groups[i - 1] for i in range(1, len(groups))] or [n]
        tiles = sum((c * hw * hw + mt - 1) // mt for c in counts)
        cout_s = y.shape[-1]
        stats = torch.full(((4 if d.nphases > 1 else 1) * tiles, 2, cout_s), float("nan"), device="cuda")
        d.stats, d.ldstat = stats.data_ptr(), cout_s
        K.gemm_nt(d)
        torch.cuda.synchronize()
        assert torch.isfinite(y.float()).all() and torch.isfinite(stats[:, :, :cout]).all()
        outs[mode], sums[mode] = y.clone(), stats[:, :, :cout].double().sum(0).cpu()
    assert torch.equal(outs["patch"], outs["stream_ct_outer"])
    scale = sums["stream"].abs().max().item()
    assert (sums["patch"] - sums["stream_ct_outer"]).abs().max().item() < 1e-5 * scale
    assert (sums["patch"] - sums["stream"]).abs().max().item() < 1e-4 * scale
    diff = (outs["patch"].float() - outs["stream"].float()).abs().max().item()
    assert diff <= 2e-2 * outs["stream"].float().abs().max().item()       # one bf16 ulp of the largest output
    assert float(outs["patch"][..., cout:].abs().max()) == 0.0 if outs["patch"].shape[-1] > cout else True


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,c", [(5, 24), (60, 992)])
def test_logit_head_fused_matches_layer_path(n, c, dtype):
    """The critics' logit layer (Conv2d(C, 1, 4, 4) + Sigmoid, spectral-normed, biased; reference model.py:79-80) as the fused
    launches of csrc/head.hip against the general LayerFn path it replaces, on the three calls of a critic update sharing one
    launch (real: n rows, wrong: n - 1, fake: n; one power iteration and sigma per call): probabilities, dX, and the weight /
    bias gradients including the spectral-norm rank-1 terms; and against float64 torch for the forward."""
    import copy
    import torch.nn as nn
    from cpcsv import modules as M, runtime
    from cpcsv.runtime import row_groups
    runtime.set_compute_dtype(dtype)
    torch.manual_seed(n * 31 + c)
    cs = (c + 7) // 8 * 8
    rows = 3 * n - 1
    xs = torch.zeros(rows, 4, 4, cs)
    xs[..., :c] = torch.randn(rows, 4, 4, c)
    dy = torch.randn(rows)
    base = M.Conv2d(c, 1, 4, 4, 0, bias=True, spectral=True)
    with torch.no_grad():
        base.weight_orig.normal_(0, 0.05)
        base.bias.fill_(0.1)
    outs = {}
    keep = M._LOGIT_HEAD
    try:
        for mode in (True, False):
            M._LOGIT_HEAD = mode
            seq = M.FusedSequential(copy.deepcopy(base), nn.Sigmoid(), head_last=True).cuda()
            x = xs.to(runtime.tdtype()).cuda().requires_grad_()
            with row_groups((n, n - 1, n)):
                p = seq(x).view(-1)
            assert seq._plan()[0]._by_hw[(4, 4)].logit_head == mode
            p.backward(dy.cuda())
            torch.cuda.synchronize()
            conv = seq[0]
            outs[mode] = (p.detach().float().cpu(), x.grad.float().cpu(), conv.weight_orig.grad.float().cpu(), conv.bias.grad.float().cpu(),
                          conv.weight_u.cpu().clone(), conv.weight_v.cpu().clone())
    finally:
        M._LOGIT_HEAD = keep
    (pf, dxf, dwf, dbf, uf, vf), (pl, dxl, dwl, dbl, ul, vl) = outs[True], outs[False]
    assert torch.allclose(uf, ul) and torch.allclose(vf, vl, rtol=1e-4, atol=1e-6)     # three power iterations each (atomic sums)
    tol = 2e-5 if dtype == "fp32" else 2e-2
    rel = lambda a, b: ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
    assert rel(pf, pl) < tol and rel(dxf, dxl) < tol and rel(dwf, dwl) < tol and rel(dbf, dbl) < tol, (rel(pf, pl), rel(dxf, dxl), rel(dwf, dwl), rel(dbf, dbl))
    # forward against float64: three iterations from the stored u, sigma after each one scales that call's rows
    w = base.weight_orig.detach().double().reshape(1, -1)
    u, v = base.weight_u.double(), base.weight_v.double()
    xd = (xs.to(runtime.tdtype()).double())[..., :c].permute(0, 3, 1, 2).reshape(rows, -1)          # (c, y, x) order like the master
    wd = base.weight_orig.detach().to(runtime.tdtype()).double().reshape(1, -1) if dtype == "bf16" else w
    want, r0 = [], 0
    for cnt in (n, n - 1, n):
        v = torch.nn.functional.normalize(w.t() @ u, dim=0, eps=1e-12)
        u = torch.nn.functional.normalize(w @ v, dim=0, eps=1e-12)
        sigma = float(u @ (w @ v))
        want.append(torch.sigmoid(xd[r0:r0 + cnt] @ wd.t() / sigma + 0.1).view(-1))
        r0 += cnt
    assert rel(pf.double(), torch.cat(want)) < (1e-5 if dtype == "fp32" else 5e-3)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gemm_addend_and_weight_slice_stride(dtype):
    """cpcsv_gemm_desc.addend (fp32 [rows][ldadd] added after alpha, before bias / statistics / activation; also behind a split-K
    pass) and .wstride (the K slices of consecutive taps are `wstride` elements apart in B while A stores only Cs channels: the
    feature channels of a wider layer's packed weight) against float64: a 3-tap product over M rows."""
    import ctypes as C
    from cpcsv import kernels as K, _lib as L
    td = torch.float32 if dtype == "fp32" else torch.bfloat16
    dt = L.F32 if dtype == "fp32" else L.BF16
    torch.manual_seed(3)
    m, n, cs, ws, taps = 200, 72, 64, 96, 3
    a = torch.randn(m, 1, 1, cs).to(td).cuda()                       # 1x1 "maps": every tap reads the row itself
    bfull = (torch.randn(n, taps * ws) * 0.1).to(td).cuda()           # [N][taps][ws], the first cs of each slice are used
    add = torch.randn(m, 72).cuda()
    alpha = torch.tensor([0.5], device="cuda")
    bias = torch.randn(n).cuda()
    want = sum(a.double().view(m, cs).cpu() @ bfull.double().cpu().view(n, taps, ws)[:, t, :cs].t() for t in range(taps))
    want = torch.relu(want * 0.5 + add.double().cpu() + bias.double().cpu())
    for splitk in (1, 3):
        y = torch.full((m, 72), float("nan"), dtype=td, device="cuda")
        d = K.gemm_desc(a, bfull, y, dtype=dt, M=m, N=n, Cs=cs, ldb=taps * ws, ldc=72, taps=[(0, 0, t) for t in range(taps)],
                        alpha=alpha, bias=bias, act=L.ACT_RELU)
        d.wstride, d.addend, d.ldadd = ws, add.data_ptr(), 72
        if splitk > 1:
            wsb = torch.empty(splitk, m, 72, device="cuda")
            d.splitk, d.ws, d.ldws, d.ws_rows = splitk, wsb.data_ptr(), 72, m
        K.gemm_nt(d)
        torch.cuda.synchronize()
        err = (y.double().cpu() - want).abs().max().item() / want.abs().max().item()
        assert err < (1e-5 if dtype == "fp32" else 1.5e-2), (splitk, err)
    d.ldadd = 70                                                       # not a multiple of 4 / narrower than N: argument error
    assert L.load().cpcsv_gemm_nt(C.byref(d), None) == -1009


def test_small_weight_gradients_batched_in_one_launch():
    """cpcsv_dense_rows_wgrad_multi (the parked weight gradients of the text / motion encoders and GRU cells, one launch at the end
    of the generator's backward): several weights, several pieces per weight (story half + image half; the stacked GRU steps),
    with and without a bias gradient, against the one-launch-per-piece kernel - bit-identical (same order of additions) - and
    through the runtime's park / flush path incl. more weights than one launch takes."""
    from cpcsv import kernels as K, runtime
    torch.manual_seed(9)
    specs = [(248, 1784, [12, 60], True), (372, 128, [60, 12], True), (1095, 468, [60, 60], False), (9, 40, [5], True)] + \
            [(40 + i, 24 + 8 * i, [7, 64], bool(i & 1)) for i in range(18)]
    ref, got, jobs = [], [], []
    for n, kr, ms, has_b in specs:
        ldz, ldx = (n + 7) // 8 * 8, (kr + 7) // 8 * 8
        w0, b0 = torch.randn(n, kr, device="cuda"), torch.randn(n, device="cuda")
        wa, ba, wb, bb = w0.clone(), b0.clone(), w0.clone(), b0.clone()
        for m in ms:
            dz = torch.zeros(m, ldz, device="cuda")
            dz[:, :n] = torch.randn(m, n, device="cuda")
            x = torch.randn(m, ldx, device="cuda")
            K.dense_rows_wgrad(dz, x, wa, m, n, kr, ba if has_b else None)
            jobs.append((wb, bb if has_b else None, dz, x, m, n, kr))
        ref.append((wa, ba))
        got.append((wb, bb))
    runtime.defer_small_wgrads(True)
    try:
        assert runtime.small_wgrads_deferred()
        for j in jobs:
            runtime.park_small_wgrad(*j)
        runtime.flush_small_wgrads()
    finally:
        runtime.defer_small_wgrads(False)
    torch.cuda.synchronize()
    for (wa, ba), (wb, bb) in zip(ref, got):
        assert torch.equal(wa, wb) and torch.equal(ba, bb)


def test_dense_operand_copies_batched_in_one_launch():
    """cpcsv_pack_dense_many (the nine small fp32 layers of the text / motion encoders, stale once per step) against
    cpcsv_pack_weight layer by layer: bit-identical forward and transposed copies, pads zero even over poisoned buffers; through
    modules.prepack_dense the layers' cache keys move forward, so a later packs() launches nothing."""
    import ctypes as C
    from cpcsv import _lib as L, kernels as K, modules as M
    torch.manual_seed(4)
    shapes = [(248, 365), (124, 356), (372, 128), (9, 40), (1095, 468), (33, 31), (8, 8)] + [(40 + 3 * i, 24 + 5 * i) for i in range(12)]
    jobs, refs = [], []
    for cout, cin in shapes:
        cin_s, cout_s = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
        w = torch.randn(cout, cin, device="cuda")
        rf, rl = torch.full((cout, cin_s), float("nan"), device="cuda"), torch.full((cin_s, cout_s), float("nan"), device="cuda")
        K.pack_weight(w, rf, None, rl, L.F32, cout, cin, 1, 1, None, cin_s, cout_s)
        gf, gl = torch.full_like(rf, float("nan")), torch.full_like(rl, float("nan"))
        jobs.append((w, gf, gl, cout, cin, cin_s, cout_s))
        refs.append((rf, rl, gf, gl))
    K.pack_dense_many(jobs)                                   # 19 layers: two launches
    K.pack_dense_many([(jobs[0][0], None, refs[0][3].zero_(), *jobs[0][3:])])      # only the transposed copy
    torch.cuda.synchronize()
    for rf, rl, gf, gl in refs:
        assert torch.equal(rf, gf) and torch.equal(rl, gl)
    bad = L.PackList()
    bad.n = 1
    assert L.load().cpcsv_pack_dense_many(C.byref(bad), None) == -1001          # no source
    # the module path
    lins = [M.Linear(365, 248).cuda(), M.Linear(128, 372).cuda()]
    lays = [M._layer_for(l, None, L.ACT_NONE, 0, out_mode="f32pad") for l in lins]
    for lay in lays:
        lay.packs(lay.holder.master(), L.F32, "both")
    before = [tuple(t.clone() for t in lay._packs if t is not None) for lay in lays]
    with torch.no_grad():
        for l in lins:
            l.master().mul_(1.5)
    M.PACK_LOG = []
    try:
        M.prepack_dense(lays)
        assert len(M.PACK_LOG) == 2
        M.PACK_LOG[:] = []
        for lay in lays:
            lay.packs(lay.holder.master(), L.F32, "both")
        assert M.PACK_LOG == []                               # nothing was stale any more
    finally:
        M.PACK_LOG = None
    torch.cuda.synchronize()
    for lay, old in zip(lays, before):
        new = [t for t in lay._packs if t is not None]
        for a, b in zip(old, new):
            assert torch.equal(a * 1.5, b)


def test_dense_rows_rejects_narrow_operands():
    """A row stride narrower than K would make the 16-byte loads of cpcsv_dense_rows / cpcsv_gru_step_fwd walk past the row
    (out of bounds on the last one): argument error -1001, nothing launched."""
    from cpcsv import _lib as L
    lib = L.load()
    x, w, y = torch.zeros(4, 16, device="cuda"), torch.zeros(8, 16, device="cuda"), torch.full((4, 8), 7.0, device="cuda")
    P = lambda t: t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.cpcsv_dense_rows(P(x), 16, P(w), 16, P(y), 8, 4, 8, 16, None, None, 0, None, 0, None, 0, 0, st) == 0
    assert lib.cpcsv_dense_rows(P(x), 8, P(w), 16, P(y), 8, 4, 8, 16, None, None, 0, None, 0, None, 0, 0, st) == -1001      # ldx < K
    assert lib.cpcsv_dense_rows(P(x), 16, P(w), 12, P(y), 8, 4, 8, 16, None, None, 0, None, 0, None, 0, 0, st) == -1001     # ldw < K
    gi, h, whh, bhh = torch.zeros(4, 24, device="cuda"), torch.zeros(4, 8, device="cuda"), torch.zeros(24, 8, device="cuda"), torch.zeros(24, device="cuda")
    hn, gates = torch.zeros(4, 8, device="cuda"), torch.zeros(4, 32, device="cuda")
    assert lib.cpcsv_gru_step_fwd(P(gi), 24, P(h), 8, P(whh), 8, P(bhh), P(hn), P(gates), 4, 8, st) == 0
    assert lib.cpcsv_gru_step_fwd(P(gi), 24, P(h), 8, P(whh), 4, P(bhh), P(hn), P(gates), 4, 8, st) == -1001                # ldw < ldh
    assert lib.cpcsv_gru_step_fwd(P(gi), 16, P(h), 8, P(whh), 8, P(bhh), P(hn), P(gates), 4, 8, st) == -1001                # ldg < 3H
    torch.cuda.synchronize()


@pytest.mark.parametrize("t_,b,inp,hid", [(5, 12, 45, 23), (1, 60, 37, 124), (5, 13, 20, 365), (3, 70, 16, 9)])
def test_gru_sequence_matches_torch(t_, b, inp, hid):
    """M.GRUCell.sequence (one launch per step forward, two per step + one weight-gradient launch backward; more than 64 rows fall
    back to step()) against nn.GRUCell unrolled over T steps (reference model.py:320-346): states, d input, d h0, all four
    parameter gradients."""
    import torch.nn as nn
    from cpcsv import functional as F
    from cpcsv import modules as M
    from cpcsv import runtime
    runtime.set_compute_dtype("fp32")
    torch.manual_seed(t_ * 100 + b)
    tg, pg = nn.GRUCell(inp, hid), M.GRUCell(inp, hid)
    pg.load_state_dict(tg.state_dict())
    pg.to("cuda")
    x, h0 = torch.randn(t_, b, inp), torch.randn(b, hid)
    xt, ht = x.clone().requires_grad_(), h0.clone().requires_grad_()
    h, outs = ht, []
    for t in range(t_):
        h = tg(xt[t], h)
        outs.append(h)
    yt = torch.stack(outs, 0)
    dy = torch.randn_like(yt)
    yt.backward(dy)
    xp, hp = x.cuda().requires_grad_(), h0.cuda().requires_grad_()
    hs = (hid + 7) // 8 * 8
    hpad = torch.cat((hp, torch.zeros(b, hs - hid, device="cuda")), 1)
    gi = pg.input_gates(M.dense_input(xp.reshape(t_ * b, inp), dtype=pg.in_dtype())).view(t_, b, -1)
    yp = pg.sequence(gi, hpad)[:, :, :hid]
    yp.backward(dy.cuda())
    torch.cuda.synchronize()
    rel = lambda a, c: ((a.detach().float().cpu() - c.detach()).abs().max() / (c.detach().abs().max() + 1e-12)).item()
    assert rel(yp, yt) < 1e-5 and rel(xp.grad, xt.grad) < 1e-4 and rel(hp.grad, ht.grad) < 1e-4
    for k, p_ in pg.named_parameters():
        assert rel(p_.grad, dict(tg.named_parameters())[k].grad) < 1e-4, k


@pytest.mark.parametrize("off,nbytes", [(0, 0), (0, 1), (3, 5), (1, 15), (7, 16), (0, 4096), (5, 4099), (13, 1 << 20), (0, (1 << 26) + 7)])
def test_fill_zero_touches_exactly_its_range(off, nbytes):
    """cpcsv_fill_zero (this library's own launch since round 5, no hipMemsetAsync): unaligned heads and ragged tails, sizes past
    the grid-stride limit - every byte of [off, off + nbytes) zero, every byte outside untouched."""
    from cpcsv import kernels as K
    buf = torch.full((off + nbytes + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    K.fill_zero(buf[off:off + nbytes])
    torch.cuda.synchronize()
    host = buf.cpu()
    assert int(host[off:off + nbytes].max()) == 0 if nbytes else True
    assert bool((host[:off] == 0xA5).all()) and bool((host[off + nbytes:] == 0xA5).all())


def _cond_head_case(cf, e, cout, n, dtype, triplet, seed=0, factored=True):
    """D_GET_LOGITS' first layer (reference model.py:75-78,89-92: SN-conv3x3 over [features | tiled condition] + BatchNorm +
    LeakyReLU) on the product (factored form, csrc/condhead.hip, or the literal concatenated form) and in plain PyTorch fp32 with one
    CALL per reference call (miscc/utils.py:70-84: real, wrong = real[:N-1] with cond[1:], fake). Returns {name: (product, torch)}."""
    import torch.nn as nn
    from cpcsv import functional as F
    from cpcsv import modules as M
    from cpcsv import runtime
    from oracle.cpcsv_oracle.nets import SpectralConv2d
    runtime.set_compute_dtype(dtype)
    torch.manual_seed(seed)
    tnet = nn.Sequential(SpectralConv2d(cf + e, cout, 3, 1, 1, False), nn.BatchNorm2d(cout), nn.LeakyReLU(0.2))
    tnet[1].weight.data.uniform_(0.5, 1.5); tnet[1].bias.data.uniform_(-0.5, 0.5)
    pnet = M.FusedSequential(M.Conv2d(cf + e, cout, 3, 1, 1, bias=False, spectral=True), M.BatchNorm2d(cout), nn.LeakyReLU(0.2))
    feats = torch.randn((2 * n if triplet else n), cf, 4, 4)
    cond = torch.randn(n, e)
    if dtype == "bf16":
        feats, cond = feats.bfloat16().float(), cond.bfloat16().float()
        with torch.no_grad():
            for p_ in tnet.parameters():
                if p_.dim() > 1:
                    p_.copy_(p_.bfloat16().float())
    pnet.load_state_dict(tnet.state_dict(), strict=True)
    pnet.cuda()
    pnet._plan()[0].dgrad_cols = cf
    ft = feats.clone().requires_grad_()
    tile = lambda c: c.view(-1, e, 1, 1).repeat(1, 1, 4, 4)
    if triplet:
        calls = [(ft[:n], cond), (ft[:n - 1], cond[1:]), (ft[n:], cond)]
    else:
        calls = [(ft, cond)]
    yt = torch.cat([tnet(torch.cat((f, tile(c)), 1)) for f, c in calls], 0)
    dy = torch.randn_like(yt)
    yt.backward(dy)
    fp = feats.clone().cuda().requires_grad_()
    h = F.ToNhwcFn.apply(fp, runtime.tdtype())
    lay = pnet._plan()[0]
    groups = (n, n - 1, n) if triplet else None
    keep = M._COND_HEAD
    M._COND_HEAD = factored
    try:
        assert lay.cond_head_ok(h, cond.cuda(), groups) == factored
        with runtime.row_groups(groups):
            if factored:
                yp = lay(h, cond=cond.cuda())
            elif triplet:
                x = F.CondTripletFn.apply(h, cond.cuda(), cf)
                x._cpcsv_live_cols = cf
                yp = lay(x)
            else:
                x = F.CondConcatFn.apply(h, cond.cuda(), cf)
                x._cpcsv_live_cols = cf
                yp = lay(x)
        assert any(isinstance(k, tuple) and k[0] == "cond_head" for k in lay.descs) == factored
    finally:
        M._COND_HEAD = keep
    yp = F.ToPlanarFn.apply(yp, cout)
    yp.backward(dy.cuda())
    torch.cuda.synchronize()
    out = {"y": (yp, yt), "dfeat": (fp.grad, ft.grad)}
    tp = dict(tnet.named_parameters())
    for k, p_ in pnet.named_parameters():
        out["d_" + k] = (p_.grad, tp[k].grad)
    tb = dict(tnet.named_buffers())
    for k, b in pnet.state_dict().items():
        if k in tb and tb[k].dtype.is_floating_point:
            out["buf_" + k] = (b, tb[k])
    return {k: (a.detach().float().cpu(), b.detach().float()) for k, (a, b) in out.items()}


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(16, 5, 16, 4), (136, 9, 128, 6), (64, 20, 64, 3)])
@pytest.mark.parametrize("triplet", [True, False])
def test_cond_head_factored_matches_torch(shape, dtype, triplet):
    """The factored conditional head (feature conv over the distinct feature maps + per-tap condition products + the assembling
    BatchNorm kernel) against PyTorch fp32 making the reference's separate calls: outputs, feature gradients, weight_orig / BatchNorm
    gradients, running statistics, spectral-norm u / v. Same tolerances as every other fused layer."""
    cf, e, cout, n = shape
    rep = _cond_head_case(cf, e, cout, n, dtype, triplet)
    ftol, gtol = op_cases.tolerances(dtype)
    scale = max(b.abs().max().item() for k, (a, b) in rep.items() if k.startswith("d_"))
    for k, (a, b) in rep.items():
        den = max(b.abs().max().item(), 1e-2 * scale) if k.startswith("d_") else b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item() / den
        assert err < (gtol if k.startswith("d") else ftol), (shape, dtype, triplet, k, err)


@pytest.mark.parametrize("triplet", [True, False])
def test_cond_head_factored_agrees_with_the_literal_form(triplet):
    """fp32: the factored form against the literal concatenated form it replaces (same weights, same inputs; summation order differs)."""
    a = _cond_head_case(136, 9, 128, 6, "fp32", triplet, factored=True)
    b = _cond_head_case(136, 9, 128, 6, "fp32", triplet, factored=False)
    for k in a:
        err = (a[k][0] - b[k][0]).abs().max().item() / (b[k][0].abs().max().item() + 1e-12)
        assert err < 2e-4, (k, err)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_cond_head_factored_full_width(dtype):
    """cfg/final.yml widths and the benchmark's batch: 992 feature + 489 condition channels -> 992, N = 60 (179 head samples)."""
    rep = _cond_head_case(992, 489, 992, 60, dtype, True)
    ftol, gtol = op_cases.tolerances(dtype)
    scale = max(b.abs().max().item() for k, (a, b) in rep.items() if k.startswith("d_"))
    for k, (a, b) in rep.items():
        den = max(b.abs().max().item(), 1e-2 * scale) if k.startswith("d_") else b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item() / den
        assert err < (gtol if k.startswith("d") else ftol), (dtype, k, err)


def test_batch_prep_matches_the_reference_slicing():
    """cpcsv_batch_prep against the torch ops of reference trainer.py:254-264,287-288,303-304 (bit-exact: copies, a T-term mean, a compare)."""
    from cpcsv import kernels as K
    torch.manual_seed(2)
    im, st, t, td, l, d = 7, 3, 5, 20, 3, 23
    idesc, ilab, icont = torch.randn(im, d).cuda(), (torch.rand(im, l) < 0.3).float().cuda(), torch.randn(im, t, d).cuda()
    sdesc, slab = torch.randn(st, t, d).cuda(), (torch.rand(st, t, l) < 0.3).float().cuda()
    slab[0] = 0.0                                                  # a story without any label: chars = 0
    im_m, im_c, st_m, st_x, st_mean, chars = K.batch_prep(idesc, ilab, icont, sdesc, slab, td)
    torch.cuda.synchronize()
    assert torch.equal(im_m, torch.cat((idesc[:, :td], ilab), 1)) and torch.equal(im_c, icont[:, :, :td].contiguous())
    st_text = sdesc[:, :, :td]
    assert torch.equal(st_m, torch.cat((st_text, slab), 2)) and torch.equal(st_x, st_text.contiguous())
    assert torch.allclose(st_mean, st_text.mean(1), rtol=1e-6, atol=1e-7)
    assert torch.equal(chars, (slab.mean(1) > 0).float())


@pytest.mark.parametrize("onepass", [True, False])
def test_spectral_plan_iterations_match_float64(onepass):
    """SpectralPlan (cpcsv/spectral.py): the power iterations of MANY spectral-normed layers per launch triple, three rounds in a row
    (call k+1 of a layer starts from the u of call k), in the ONE-pass form (cpcsv_spectral_sigma_multi1: column slabs in registers,
    W read once per iteration) and in the two-pass form, against torch.nn.utils.spectral_norm's arithmetic in float64: per call
    sigma to 1e-5, u and v to 1e-5 of their unit length; the layers' u / v buffers end on the last call's. Shapes: the critics' real
    ones incl. the head conv's odd row length (1481 * 9 = 13329 columns), a one-row logit layer, Linear layers of the order critic,
    rows just under / over the register buckets (256 / 512)."""
    from cpcsv import modules as M, spectral as S, runtime
    assert not runtime.deterministic()
    torch.manual_seed(3)
    layers = [M.Conv2d(1481, 992, 3, 1, 1, bias=False, spectral=True), M.Conv2d(496, 992, 4, 2, 1, bias=False, spectral=True),
              M.Conv2d(124, 248, 4, 2, 1, bias=False, spectral=True), M.Conv2d(992, 1, 4, 4, 0, bias=True, spectral=True),
              M.Linear(512, 128, spectral=True), M.Linear(128, 1, spectral=True), M.Conv2d(7, 257, 3, 1, 1, bias=False, spectral=True),
              M.Conv2d(5, 513, 3, 1, 1, bias=False, spectral=True), M.Conv2d(3, 45, 3, 1, 1, bias=False, spectral=True)]
    for m in layers:
        m.cuda()
        with torch.no_grad():
            m.master().normal_(0, 0.05)
    keep = S._ONEPASS
    S._ONEPASS = onepass
    try:
        plan = S.SpectralPlan([(m, 3, 1) for m in layers])
        ref = []
        for m in layers:
            rows, cols = m._sn_shape
            w = m.master().detach().double().reshape(rows, cols).cpu()
            u, v = m.weight_u.double().cpu().clone(), m.weight_v.double().cpu().clone()
            calls = []
            for _ in range(3):
                v = torch.nn.functional.normalize(w.t() @ u, dim=0, eps=1e-12)
                u = torch.nn.functional.normalize(w @ v, dim=0, eps=1e-12)
                calls.append((float(u @ (w @ v)), u.clone(), v.clone()))
            ref.append(calls)
        plan.run("D")
        torch.cuda.synchronize()
        tab = plan._round("D", 0)
        assert (tab[8] is not None) == onepass
        for m, calls in zip(layers, ref):
            assert len(m._sn_queue) == 3
            for (sig, us, vs), (s64, u64, v64) in zip(m._sn_queue, calls):
                assert abs(float(sig[0]) - s64) < 1e-5 * abs(s64) and abs(float(sig[1]) - 1.0 / s64) < 1e-5 / abs(s64), (m._sn_shape, float(sig[0]), s64)
                assert (us.double().cpu() - u64).abs().max().item() < 1e-5 and (vs.double().cpu() - v64).abs().max().item() < 1e-5, m._sn_shape
            assert (m.weight_u.double().cpu() - calls[-1][1]).abs().max().item() < 1e-5
            assert (m.weight_v.double().cpu() - calls[-1][2]).abs().max().item() < 1e-5
            assert float(m._sn_work.abs().max()) == 0.0                # the accumulators are left zero for the next call
    finally:
        S._ONEPASS = keep
