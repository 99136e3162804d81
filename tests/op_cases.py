"""Op-level parity cases: a product FusedSequential (HIP) vs the same stack in plain PyTorch fp32 on CPU.
Used by tests/test_gpu_ops.py and tools/gpu_diag.py."""
import torch
import torch.nn as nn

from oracle.cpcsv_oracle.nets import SpectralConv2d


def _torch_and_product(spec):
    """spec: list of tuples describing layers -> (torch nn.Sequential, product FusedSequential)."""
    from cpcsv import modules as M
    t, p = [], []
    for item in spec:
        kind = item[0]
        if kind == "up":
            t.append(nn.Upsample(scale_factor=2, mode="nearest")); p.append(M.Upsample())
        elif kind == "conv":
            _, cin, cout, k, s, pad, bias, sn = item
            t.append(SpectralConv2d(cin, cout, k, s, pad, bias) if sn else nn.Conv2d(cin, cout, k, s, pad, bias=bias))
            p.append(M.Conv2d(cin, cout, k, s, pad, bias=bias, spectral=sn))
        elif kind == "lin":
            _, cin, cout, bias = item
            t.append(nn.Linear(cin, cout, bias=bias)); p.append(M.Linear(cin, cout, bias=bias))
        elif kind == "bn2":
            t.append(nn.BatchNorm2d(item[1])); p.append(M.BatchNorm2d(item[1]))
        elif kind == "bn1":
            t.append(nn.BatchNorm1d(item[1])); p.append(M.BatchNorm1d(item[1]))
        elif kind == "relu":
            t.append(nn.ReLU()); p.append(nn.ReLU())
        elif kind == "lrelu":
            t.append(nn.LeakyReLU(0.2)); p.append(nn.LeakyReLU(0.2))
        elif kind == "tanh":
            t.append(nn.Tanh()); p.append(nn.Tanh())
        elif kind == "sigmoid":
            t.append(nn.Sigmoid()); p.append(nn.Sigmoid())
    return t, p


CASES = {
    # name: (spec, input shape NCHW or (B,K), kwargs for FusedSequential)
    "conv3x3": ([("conv", 16, 24, 3, 1, 1, False, False)], (3, 16, 8, 8), {}),
    "conv3x3_odd_tiles": ([("conv", 160, 136, 3, 1, 1, False, False)], (5, 160, 12, 12), {}),
    "upblock": ([("up",), ("conv", 32, 16, 3, 1, 1, False, False), ("bn2", 16), ("relu",)], (3, 32, 4, 4), {}),
    "upblock_wide": ([("up",), ("conv", 128, 64, 3, 1, 1, False, False), ("bn2", 64), ("relu",)], (4, 128, 8, 8), {}),
    # the same blocks with the sub-pixel form switched off: the direct 9-tap gather with the upsample folded in
    "upblock_direct": ([("up",), ("conv", 32, 16, 3, 1, 1, False, False), ("bn2", 16), ("relu",)], (3, 32, 4, 4), {}),
    "upblock_wide_direct": ([("up",), ("conv", 128, 64, 3, 1, 1, False, False), ("bn2", 64), ("relu",)], (4, 128, 8, 8), {}),
    "upblock_odd": ([("up",), ("conv", 40, 24, 3, 1, 1, False, False), ("bn2", 24), ("relu",)], (3, 40, 5, 7), {}),
    "d_enc0": ([("conv", 3, 12, 4, 2, 1, False, False), ("lrelu",)], (4, 3, 16, 16), {}),
    "d_enc_sn_bn": ([("conv", 12, 24, 4, 2, 1, False, True), ("bn2", 24), ("lrelu",)], (4, 12, 16, 16), {}),
    "d_enc_sn_first": ([("conv", 3, 8, 4, 2, 1, False, True), ("lrelu",)], (4, 3, 16, 16), {}),
    "downblock": ([("conv", 8, 16, 3, 2, 1, True, False), ("bn2", 16), ("relu",)], (3, 8, 16, 16), {}),
    "img_tanh": ([("conv", 8, 3, 3, 1, 1, False, False), ("tanh",)], (2, 8, 16, 16), {}),
    # the streaming kernels of csrc/thin.hip (bf16 only; fp32 mode runs the same cases through the gather-GEMM):
    # StoryGAN.img 128->3 / img_seg 64->1 (+tanh) and the critics' first conv 3|1->124 k4 s2 p1 (+LeakyReLU, SN in D_STY)
    "thin_img": ([("conv", 128, 3, 3, 1, 1, False, False), ("tanh",)], (2, 128, 32, 32), {}),
    "thin_img_ragged_rows": ([("conv", 128, 3, 3, 1, 1, False, False), ("tanh",)], (1, 128, 5, 32), {}),
    "thin_seg": ([("conv", 64, 1, 3, 1, 1, False, False), ("tanh",)], (3, 64, 32, 32), {}),
    "thin_enc0": ([("conv", 3, 124, 4, 2, 1, False, False), ("lrelu",)], (3, 3, 32, 32), {}),
    "thin_enc0_seg": ([("conv", 1, 124, 4, 2, 1, False, False), ("lrelu",)], (2, 1, 32, 64), {}),
    "thin_enc0_sn": ([("conv", 3, 124, 4, 2, 1, False, True), ("lrelu",)], (3, 3, 32, 32), {}),
    "thin_enc0_sn_wide": ([("conv", 3, 124, 4, 2, 1, False, True), ("lrelu",)], (2, 3, 64, 64), {}),   # 64-wide: streaming dgrad / wgrad too
    "seg_tanh": ([("conv", 4, 1, 3, 1, 1, False, False), ("tanh",)], (2, 4, 16, 16), {}),
    "head_logits": ([("conv", 24, 16, 3, 1, 1, False, True), ("bn2", 16), ("lrelu",),
                     ("conv", 16, 1, 4, 4, 0, True, True), ("sigmoid",)], (5, 24, 4, 4), {"head_last": True}),
    "fc_bn_relu": ([("lin", 37, 96, False), ("bn1", 96), ("relu",)], (6, 37), {"out_mode": "f32"}),
    "fc_bias": ([("lin", 45, 22, True), ("relu",)], (7, 45), {"out_mode": "f32"}),
    "fc_bn_tanh": ([("lin", 23, 372, True), ("bn1", 372), ("tanh",)], (5, 23), {"out_mode": "f32"}),
    # few-tile / long-K shapes: exercise the split-K path (fp32 workspace + epilogue pass) and the batched
    # transposed-conv phases, as the wide 4x4 / 8x8 layers of the real model do
    "conv3x3_splitk": ([("conv", 256, 40, 3, 1, 1, False, False)], (3, 256, 4, 4), {}),
    "upblock_splitk": ([("up",), ("conv", 192, 80, 3, 1, 1, False, False), ("bn2", 80), ("relu",)], (2, 192, 2, 2), {}),
    "d_enc_splitk": ([("conv", 64, 256, 4, 2, 1, False, True), ("bn2", 256), ("lrelu",)], (2, 64, 8, 8), {}),
    "downblock_splitk": ([("conv", 128, 256, 3, 2, 1, True, False), ("bn2", 256), ("relu",)], (2, 128, 8, 8), {}),
    "fc_splitk": ([("lin", 1200, 40, True), ("relu",)], (6, 1200), {"out_mode": "f32"}),
    "head_splitk": ([("conv", 136, 128, 3, 1, 1, False, True), ("bn2", 128), ("lrelu",),
                     ("conv", 128, 1, 4, 4, 0, True, True), ("sigmoid",)], (6, 136, 4, 4), {"head_last": True}),
}


# the layers of the BASELINE configuration at their real widths (cfg/final.yml: GF_DIM 256, DF_DIM 124, 64x64 images;
# 12-60 images so that the CPU fp32 reference of one layer finishes in a few seconds) - tests/test_gpu_fullsize.py
FULL_CASES = {
    "full_up2": ([("up",), ("conv", 1024, 512, 3, 1, 1, False, False), ("bn2", 512), ("relu",)], (12, 1024, 8, 8), {}),
    "full_up4": ([("up",), ("conv", 256, 128, 3, 1, 1, False, False), ("bn2", 128), ("relu",)], (12, 256, 32, 32), {}),
    "full_d_enc2": ([("conv", 124, 248, 4, 2, 1, False, True), ("bn2", 248), ("lrelu",)], (60, 124, 32, 32), {}),
    "full_d_enc4": ([("conv", 496, 992, 4, 2, 1, False, True), ("bn2", 992), ("lrelu",)], (60, 496, 8, 8), {}),
    "full_head": ([("conv", 1481, 992, 3, 1, 1, False, True), ("bn2", 992), ("lrelu",),
                   ("conv", 992, 1, 4, 4, 0, True, True), ("sigmoid",)], (60, 1481, 4, 4), {"head_last": True}),
    "full_fc": ([("lin", 616, 32768, False), ("bn1", 32768), ("relu",)], (60, 616), {"out_mode": "f32"}),
    "full_img": ([("conv", 128, 3, 3, 1, 1, False, False), ("tanh",)], (12, 128, 64, 64), {}),
    "full_img_seg": ([("conv", 64, 1, 3, 1, 1, False, False), ("tanh",)], (12, 64, 64, 64), {}),
    "full_d_enc0": ([("conv", 3, 124, 4, 2, 1, False, False), ("lrelu",)], (24, 3, 64, 64), {}),
    # DF_DIM 113..120: stored stride 120, NOT served by the 128-stride streaming kernels (must take the gather-GEMM)
    "full_d_enc0_c116": ([("conv", 3, 116, 4, 2, 1, False, False), ("lrelu",)], (8, 3, 64, 64), {}),
    # cascade_model.downBlock at cfg/final.yml widths (ngf_seg 1024: 64->128 on 64x64 ... 512->1024 on 8x8) and presample
    "full_down1": ([("conv", 64, 128, 3, 2, 1, True, False), ("bn2", 128), ("relu",)], (6, 64, 64, 64), {}),
    "full_down4": ([("conv", 512, 1024, 3, 2, 1, True, False), ("bn2", 1024), ("relu",)], (12, 512, 8, 8), {}),
    "full_presample": ([("conv", 1, 64, 3, 1, 1, False, False), ("bn2", 64), ("relu",)], (6, 1, 64, 64), {}),
}


def run_case(name, dtype, device="cuda", seed=0, raw=False, groups=None):
    """Returns dict of max relative errors (vs per-tensor max magnitude).
    groups=(n0, n1, ...): the torch side makes one CALL per group of rows (own BatchNorm statistics, one spectral-norm
    iteration and one running-statistics update per call, gradients accumulating over the calls - what the reference does
    with a critic's real / fake batches or a generator's story / image halves); the product side runs all groups in ONE
    set of launches (cpcsv.runtime.row_groups)."""
    from cpcsv import functional as F
    from cpcsv import modules as M
    from cpcsv import runtime
    runtime.set_compute_dtype(dtype)
    runtime.set_subpixel(not name.endswith("_direct"))
    # the *_splitk cases are small stand-ins for the long-K layers: lower the planner's thresholds so that they
    # really take the slab + epilogue path (the production thresholds only split >= 32 K tiles)
    from cpcsv import kernels as K
    K._SPLIT_MIN_NK, K._SPLIT_MINK = (8, 4) if name.endswith("_splitk") else (32, 16)
    spec, shape, kw = (FULL_CASES if name.startswith("full_") else CASES)[name]
    torch.manual_seed(seed)
    t_layers, p_layers = _torch_and_product(spec)
    tnet = nn.Sequential(*t_layers)
    for m in tnet.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.5, 0.5)
    pnet = M.FusedSequential(*p_layers, **kw)
    pnet.load_state_dict(tnet.state_dict(), strict=True)
    pnet.to(device)
    x = torch.randn(*shape)
    if dtype == "bf16":   # isolate kernel correctness from input/weight rounding
        x = x.bfloat16().float()
        with torch.no_grad():
            for p_ in tnet.parameters():
                if p_.dim() > 1:
                    p_.copy_(p_.bfloat16().float())
        pnet.load_state_dict(tnet.state_dict(), strict=True)
    xt = x.clone().requires_grad_()
    if groups is None:
        yt = tnet(xt)
    else:
        assert sum(groups) == shape[0]
        yt = torch.cat([tnet(part) for part in torch.split(xt, list(groups), 0)], 0)
    dy = torch.randn_like(yt)
    yt.backward(dy)
    xp = x.clone().to(device).requires_grad_()
    conv_in = len(shape) == 4
    h = F.ToNhwcFn.apply(xp, runtime.tdtype()) if conv_in else xp
    with runtime.row_groups(groups):
        yp = pnet(h)
    if conv_in and not kw.get("head_last"):
        yp = F.ToPlanarFn.apply(yp, yt.shape[1])
    yp = yp.reshape(yt.shape)
    yp.backward(dy.to(device))
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.detach().float().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12)).item()
    rep = {"y": rel(yp, yt), "dx": rel(xp.grad, xt.grad)}
    tp = dict(tnet.named_parameters())
    # a bias feeding a BatchNorm has an exactly-zero true gradient (round-off on both sides): errors are
    # measured against the layer-wide gradient scale, never against a tensor's own ~1e-7 magnitude
    floor = 1e-2 * max(g.grad.abs().max().item() for g in tp.values())
    relf = lambda a, b: ((a.detach().float().cpu() - b.detach()).abs().max() / max(b.detach().abs().max().item(), floor)).item()
    for k, p_ in pnet.named_parameters():
        rep["d_" + k] = relf(p_.grad, tp[k].grad)
    tb = dict(tnet.named_buffers())
    for k, b in pnet.state_dict().items():
        if k in tb and tb[k].dtype.is_floating_point:
            rep["buf_" + k] = rel(b, tb[k])
    if raw:      # the product's own tensors (comparisons between two launch paths) and which path each layer took
        tensors = {"y": yp.detach().float().cpu(), "dx": xp.grad.detach().float().cpu()}
        tensors.update({"d_" + k: p_.grad.detach().float().cpu() for k, p_ in pnet.named_parameters()})
        kinds = [v for lay in pnet._plan() for k, v in getattr(lay, "descs", {}).items() if isinstance(k, tuple) and k[0] == "thin"]
        return rep, tensors, kinds
    return rep


# name -> (case it reuses, batch, groups): several passes of one layer in ONE set of launches
GROUP_CASES = {
    "g_tower": ("d_enc_sn_bn", 7, (4, 3)),                  # critic tower layer: real | fake, uneven
    "g_tower_splitk": ("d_enc_splitk", 5, (2, 3)),
    "g_upblock": ("upblock_wide", 9, (5, 4)),               # generator: story half | image half, sub-pixel form
    "g_upblock_direct": ("upblock_wide_direct", 9, (5, 4)),
    "g_fc": ("fc_bn_relu", 11, (5, 6)),                          # dense + BatchNorm1d
    "g_head3": ("head_logits", 11, (4, 3, 4)),                     # conditional head: real | wrong (N-1) | fake
    "g_first_sn": ("d_enc_sn_first", 6, (3, 3)),            # spectral norm without BatchNorm: per-pass scale
    "g_conv_plain": ("conv3x3", 5, (2, 3)),
    "g_upblock_splitk": ("upblock_splitk", 5, (2, 3)),      # sub-pixel form + split-K epilogue pass
    "g_head_splitk": ("head_splitk", 11, (4, 3, 4)),
    "g_thin_first_sn": ("thin_enc0_sn_wide", 4, (2, 2)),    # bf16: streaming first conv, one launch per pass
}


def run_group_case(name, dtype, **kw):
    base, batch, groups = GROUP_CASES[name]
    spec, shape, ckw = CASES[base]
    CASES["__g"] = (spec, (batch,) + tuple(shape[1:]), ckw)
    try:
        nm = "__g" + ("_splitk" if base.endswith("_splitk") else "") + ("_direct" if base.endswith("_direct") else "")
        CASES[nm] = CASES["__g"]
        return run_case(nm, dtype, groups=groups, **kw)
    finally:
        CASES.pop("__g", None)
        CASES.pop(nm, None)


def tolerances(dtype):
    # (forward/buffers, grads). bf16 gradients through BN+ReLU on these tiny maps flip a few ReLU masks
    # (|z| below the bf16 step), which moves small-sample sums by several percent of their max.
    return (2e-4, 2e-3) if dtype == "fp32" else (2e-2, 0.25)
