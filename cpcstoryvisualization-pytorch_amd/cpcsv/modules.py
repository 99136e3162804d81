"""nn.Module building blocks whose forward/backward run on the HIP kernels.

They keep the reference's parameter names (so `state_dict()` keys equal the reference's, SURVEY
§8(f) F2) and class names containing 'Conv' / 'BatchNorm' / 'Linear' (so the reference's
`weights_init`, miscc/utils.py:191-201, dispatches the same way), but hold no torch.nn compute:
`FusedSequential` turns  [Upsample] -> Conv|Linear -> [BatchNorm] -> [activation]  runs into single
`LayerFn` autograd nodes (one gather-GEMM + BN kernels).
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as F
from . import kernels as K
from .runtime import current_groups, dcode, pad8, tdtype
from .runtime import subpixel as runtime_subpixel

_ACT_OF = {nn.ReLU: L.ACT_RELU, nn.LeakyReLU: L.ACT_LRELU, nn.Tanh: L.ACT_TANH, nn.Sigmoid: L.ACT_SIGMOID}


# ------------------------------------------------------------------------------------------------
# parameter holders
# ------------------------------------------------------------------------------------------------
class Conv2d(nn.Module):
    """Parameters of nn.Conv2d (model.py:16-22,79,499-520), optionally spectral-normalised with the
    old hook API's names: weight_orig / weight_u / weight_v (model.py:5,19)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, bias=True, spectral=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad, self.spectral = cin, cout, k, stride, pad, spectral
        wshape = self._weight_shape()
        fan_in = int(torch.Size(wshape[1:]).numel())
        self._sn_shape, self._sn_work = (cout, fan_in), None
        w = torch.empty(*wshape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(fan_in)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        if spectral:
            self.weight_orig = nn.Parameter(w)
            self.register_buffer("weight_u", nn.functional.normalize(torch.randn(cout), dim=0, eps=1e-12))
            self.register_buffer("weight_v", nn.functional.normalize(torch.randn(fan_in), dim=0, eps=1e-12))
        else:
            self.weight = nn.Parameter(w)
        self._layers = {}

    def _weight_shape(self):
        return (self.cout, self.cin, self.k, self.k)

    def __getattr__(self, name):
        if name == "weight" and "weight_orig" in self._parameters:
            return self._parameters["weight_orig"]     # what weights_init touches (shared storage in the reference)
        return super().__getattr__(name)

    def master(self):
        return self._parameters["weight_orig"] if self.spectral else self._parameters["weight"]

    def spectral_state(self):
        """One power iteration (train mode, also under no_grad) -> (sigma[2], u, v) for this call."""
        if not self.spectral:
            return None, None, None
        w = self.master()
        rows, cols = self._sn_shape
        need_uv = torch.is_grad_enabled() and w.requires_grad     # u v^T enters dL/dW_orig
        queue = self.__dict__.get("_sn_queue")
        if queue and self.training:
            # this call's iteration was already run by the step's SpectralPlan (cpcsv/spectral.py), in call order
            sig, us, vs = queue.pop(0)
            return (sig, us, vs) if need_uv else (sig, None, None)
        work = self._sn_work
        if work is None or work.device != w.device:
            # accumulators + tickets of the two-pass power iteration: zero once, every call leaves them zero
            work = self._sn_work = torch.zeros(rows + cols + 2, dtype=torch.float32, device=w.device)
        out = torch.empty(2 + rows + cols if need_uv else 2, dtype=torch.float32, device=w.device)
        with torch.no_grad():
            K.spectral_sigma(w, self.weight_u, self.weight_v, out, work, rows, cols, self.training, need_uv)
        if need_uv:
            return out[:2], out[2:2 + rows], out[2 + rows:]
        return out, None, None

    def forward(self, x):
        """Stand-alone 3x3 conv on an internal NHWC tensor (seg_c / seg_c1, model.py:278-279,383,387)."""
        return _layer_for(self, None, L.ACT_NONE, 0)(x)


class HeadConv2d(Conv2d):
    """A Conv2d whose window spans the whole map (cate_classify, model.py:520; called stand-alone at
    miscc/utils.py:105,150 on the NCHW-shaped feature tensor). Returns fp32 (N, Cout, 1, 1)."""

    def forward(self, x):
        return _layer_for(self, None, L.ACT_NONE, 0, head=True)(x)


class Conv3d(Conv2d):
    """Parameters of nn.Conv3d as the order critic uses it (VideoEncoder, reference model.py:18-28,117-147): the kernel is
    either spatial (1, k, k) or temporal (kt, 1, 1), always spectral-normed, never biased. The 5-D master
    [Cout][Cin][kt][kh][kw] has the [Cout][Cin][taps] memory layout the pack / unpack kernels expect. `k3`, `s3`, `p3`
    are the reference's 3-tuples; `self.k/stride/pad` are the 2-D (h, w) pairs of the conv it runs as: spatial convs on
    the frames [B*T][H][W], temporal convs on [B][T][H*W] with a (kt, 1) kernel."""

    def __init__(self, cin, cout, k3, s3, p3, spectral=True):
        self.k3, self.s3, self.p3 = tuple(k3), tuple(s3), tuple(p3)
        self.temporal = self.k3[1] == 1 and self.k3[2] == 1 and self.k3[0] > 1
        if self.temporal:
            k, st, pd = (self.k3[0], 1), (self.s3[0], 1), (self.p3[0], 0)
        else:
            k, st, pd = (self.k3[1], self.k3[2]), (self.s3[1], self.s3[2]), (self.p3[1], self.p3[2])
        super().__init__(cin, cout, k, st, pd, bias=False, spectral=spectral)

    def _weight_shape(self):
        return (self.cout, self.cin) + self.k3


class Linear(nn.Module):
    """Parameters of nn.Linear (model.py:44,251,255,261,286,303,307); spectral=True adds the old-hook spectral-norm
    names weight_orig / weight_u / weight_v (the order critic's detector, model.py:156-160)."""

    def __init__(self, cin, cout, bias=True, spectral=False):
        super().__init__()
        self.cin, self.cout, self.spectral = cin, cout, spectral
        self._sn_shape, self._sn_work = (cout, cin), None
        w = torch.empty(cout, cin)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        if spectral:
            self.weight_orig = nn.Parameter(w)
            self.register_buffer("weight_u", nn.functional.normalize(torch.randn(cout), dim=0, eps=1e-12))
            self.register_buffer("weight_v", nn.functional.normalize(torch.randn(cin), dim=0, eps=1e-12))
        else:
            self.weight = nn.Parameter(w)
        if bias:
            bound = 1.0 / math.sqrt(cin)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._layers = {}

    def __getattr__(self, name):
        if name == "weight" and "weight_orig" in self._parameters:
            return self._parameters["weight_orig"]
        return super().__getattr__(name)

    def master(self):
        return self._parameters["weight_orig"] if self.spectral else self._parameters["weight"]

    spectral_state = Conv2d.spectral_state


class _BatchNorm(nn.Module):
    """nn.BatchNorm1d/2d parameters and running buffers (eps 1e-5, momentum 0.1, model.py:32)."""

    def __init__(self, nf, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = nf, eps, momentum
        self.weight = nn.Parameter(torch.ones(nf))
        self.bias = nn.Parameter(torch.zeros(nf))
        self.register_buffer("running_mean", torch.zeros(nf))
        self.register_buffer("running_var", torch.ones(nf))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending = 0

    def note_batch(self):
        self._pending += 1      # folded into the buffer lazily: no per-call device op

    def _flush(self):
        if self._pending:
            self.num_batches_tracked += self._pending
            self._pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *a, **k):
        self._pending = 0
        super()._load_from_state_dict(*a, **k)


class BatchNorm1d(_BatchNorm):
    pass


class BatchNorm2d(_BatchNorm):
    pass


class BatchNorm3d(_BatchNorm):
    """nn.BatchNorm3d over (B, T, H, W) per channel = statistics over all rows of the [B*T*H*W][C] activation."""


class Upsample(nn.Module):
    """nn.Upsample(scale_factor=2, mode='nearest') marker (model.py:29); folded into the next conv's gather."""

    def forward(self, x):
        raise RuntimeError("Upsample is fused into the following conv; call the enclosing FusedSequential")


# ------------------------------------------------------------------------------------------------
# kernel-side view of one fused layer
# ------------------------------------------------------------------------------------------------
_F32_DENSE_MAX = 1 << 21
_GRU_SEQ = os.environ.get("CPCSV_GRU_SEQ", "1") != "0"        # GRUCell.sequence: fused recurrence (A/B switch)
_LOGIT_HEAD = os.environ.get("CPCSV_LOGIT_HEAD", "1") != "0"  # the critics' 1-output head conv as three fused launches (A/B switch)
_COND_HEAD = os.environ.get("CPCSV_COND_HEAD", "1") != "0"    # D_GET_LOGITS' 3x3 conv in factored form (A/B switch, bit-for-bit test of the old path)
PACK_LOG = None        # list while trainer.py captures a sub-graph (see GANTrainer._nograd_fakes), else None
UPDATE_LOG = None      # list while a sub-graph is captured: layers whose fused optimiser launch sits inside its backward
TERM_LOG = None        # list while a sub-graph is captured: spectral-norm terms its backward leaves for the deferred update
USE_LOG = None         # list while a sub-graph is captured: EVERY operand set the captured kernels read (graphs.py)


class KernelLayer:
    """Static description + operand caches of one conv/linear(+BN)(+act) node."""

    def __init__(self, holder, bn, act, up, kind, cin, cout, taps, slices, tapmap, geom, out_mode, name):
        self.holder, self.bn, self.act, self.kind = holder, bn, act, kind
        self.cin, self.cout, self.taps, self.slices, self.tapmap, self.geom = cin, cout, taps, slices, tapmap, geom
        self.cin_s = pad8(cin)
        self.k_stored = self.cin_s if kind == "conv" else slices * self.cin_s
        self.out_mode = out_mode          # 'T' | 'f32' | 'f32pad'
        # nearest-x2 upsample + 3x3 conv runs in its sub-pixel form (cpcsv.functional SUB_*): 16 summed-tap slices
        self.subpixel = bool(kind == "conv" and geom is not None and geom.up == 1 and geom.k == 3 and geom.s == 1
                             and geom.p == 1 and runtime_subpixel())
        if self.subpixel:
            self.slices = 16
        self.out_f32 = out_mode != "T"
        # Small dense layers (the text / motion encoders: CA_NET, m_net, c_net, filter_net, image_net, both GRU cells,
        # <= 2 M weights each, ~0.1 % of the step's FLOPs) always multiply in exact fp32: they sit behind BatchNorm1d
        # layers over as few as ST rows whose backward amplifies operand rounding by orders of magnitude (measured: the
        # fp32 reference's own gradients of these layers are only good to a few percent against fp64).
        self.compute_f32 = bool(kind == "dense" and tapmap is None and cin * cout <= _F32_DENSE_MAX)
        self.name = name
        self._packs = None
        self._pack_bufs = {}
        self._key = {}
        self._g = None
        self.fused, self.fused_terms, self.fused_dt = False, [], None     # deferred update (cpcsv.optim.FusedAdam.attach_layer)
        self.fused_opt, self.fused_seen, self.fused_expected, self.fused_updated = None, 0, 0, False
        self.fused_stash = None       # first pass of a two-pass layer, waiting for the second to share its wgrad launch
        self.descs = {}          # cached C descriptors per (pass, input shape, dtype)

    @staticmethod
    def pack_key(weight, dt):
        # _version catches torch-side in-place updates; _cpcsv_epoch is bumped by FusedAdam, whose kernel
        # writes through raw pointers and is invisible to torch's version counter
        return (weight.data_ptr(), weight._version, getattr(weight, "_cpcsv_epoch", 0), dt)

    def packs(self, weight, dt, which="both"):
        """(fwd, bwd, lin) operand copies of the master weight, rebuilt if it changed. `which`: "fwd" (forward and
        weight-gradient layout), "bwd" (the data-gradient layouts bwd/lin) or "both": the forward pass only needs the
        first, so the transposing pack of the second stays off its critical path."""
        key = self.pack_key(weight, dt)
        want = ("fwd", "bwd") if which == "both" else (which,)
        if USE_LOG is not None:
            USE_LOG.append((self, weight, dt, want))
        stale = [w for w in want if self._key.get(w) != key]
        if stale:
            if PACK_LOG is not None:                # a graph capture wants to know which layers repack inside it
                PACK_LOG.append((self, weight, dt, tuple(stale)))
            dev = weight.device
            td = torch.bfloat16 if dt == L.BF16 else torch.float32
            cout_s = pad8(self.cout)
            bufs = self._pack_bufs.get((dt, dev))
            if bufs is None:
                # allocated ONCE and re-written in place by every repack: a captured HIP graph bakes these addresses
                # into its GEMM nodes, so a repack inside the graph must land where earlier nodes of the NEXT replay read
                fwd = torch.empty(self.cout, self.slices * self.cin_s, dtype=td, device=dev)
                # (rows up to the STORED input width, zeros beyond cin: the streaming data-gradient kernel of the critics' first
                # conv - csrc/thin.hip thin4x4s2_dgrad_kernel - reads all 8 stored-channel rows of a 1- or 3-channel layer)
                bwd = torch.zeros(self.cin_s, self.slices * cout_s, dtype=td, device=dev) if self.kind == "conv" else None
                lin = torch.empty(self.slices * self.cin_s, cout_s, dtype=td, device=dev) if self.kind == "dense" else None
                if F._POISON:
                    for b in (fwd, bwd, lin):
                        if b is not None:
                            b.fill_(float("nan"))
                    if bwd is not None:
                        bwd[self.cin:].zero_()
                bufs = self._pack_bufs[(dt, dev)] = (fwd, bwd, lin)
            fwd, bwd, lin = bufs
            do_f, do_b = "fwd" in stale, "bwd" in stale
            if do_f or bwd is not None or lin is not None:
                with torch.no_grad():
                    if self.subpixel:
                        K.pack_weight_sum(weight, fwd if do_f else None, bwd if do_b else None, dt, self.cout, self.cin, self.taps,
                                          16, F.SUB_MASKS, self.cin_s, cout_s)
                    else:
                        K.pack_weight(weight, fwd if do_f else None, bwd if do_b else None, lin if do_b else None, dt, self.cout,
                                      self.cin, self.taps, self.slices, self.tapmap, self.cin_s, cout_s)
            self._packs = (fwd, bwd, lin)
            for w in stale:
                self._key[w] = key
        return self._packs

    def _dense_job(self, weight, parts):
        """This layer's entry of a prepack_dense launch - (w, fwd, lin, cout, cin, cin_s, cout_s), the stale parts, the key - or None:
        only plain fp32 dense layers, only parts whose buffers exist and are stale."""
        if not (self.kind == "dense" and self.compute_f32 and self.tapmap is None and self.slices == 1):
            return None
        key = self.pack_key(weight, L.F32)
        bufs = self._pack_bufs.get((L.F32, weight.device))
        if bufs is None:
            return None                                     # never packed yet: the first packs() call allocates
        stale = tuple(w for w in parts if self._key.get(w) != key)
        if not stale:
            return None
        fwd, _, lin = bufs
        return ((weight, fwd if "fwd" in stale else None, lin if "bwd" in stale else None, self.cout, self.cin, self.cin_s, pad8(self.cout)),
                stale, key)

    def mark_packed(self, weight, dt, parts):
        """A graph replay rebuilt `parts` from the current weights: move their cache keys forward."""
        key = self.pack_key(weight, dt)
        for w in parts:
            self._key[w] = key

    def wgrad_buffer(self, dev):
        if self._g is None or self._g.device != dev:
            self._g = torch.zeros(self.cout, self.slices * self.cin_s, dtype=torch.float32, device=dev)
        return self._g

    def in_dtype(self):
        return torch.float32 if self.compute_f32 else tdtype()

    def cond_head_ok(self, x, cond, groups):
        """Can this layer run D_GET_LOGITS' conv in its factored form (functional.LayerFn._cond_forward) on these inputs?"""
        g = self.geom
        if not (_COND_HEAD and self.kind == "conv" and g is not None and (g.k, g.s, g.p, g.up) == (3, 1, 1, 0) and self.bn is not None
                and self.bn.training and self.act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU) and x.is_cuda and x.dim() == 4
                and tuple(x.shape[1:3]) == (4, 4) and self.cout % 8 == 0 and x.shape[3] % 8 == 0 and self.holder.bias is None):
            return False
        cf = x.shape[3]
        if self.cin - cf != cond.shape[1] or getattr(self, "dgrad_cols", cf) != cf or x.dtype != tdtype():
            return False
        if groups is not None:
            n = int(groups[0])
            if tuple(int(c) for c in groups) != (n, n - 1, n) or x.shape[0] != 2 * n or cond.shape[0] != n:
                return False
            return 3 * n - 1 <= K.cond_head_max_samples()
        return cond.shape[0] == x.shape[0] and x.shape[0] <= K.cond_head_max_samples()

    def __call__(self, x, cond=None):
        h = self.holder
        if self.kind == "dense":
            if x.dim() == 4:
                x = x.contiguous().view(x.shape[0], -1)             # flattened NHWC == slices of Cin_s
            elif x.shape[1] != self.k_stored or x.dtype != self.in_dtype():
                x = dense_input(x, dtype=self.in_dtype())           # fp32 [B,K] -> padded, this layer's operand dtype
        groups = current_groups()
        if groups is not None and len(groups) < 2:
            groups = None
        if groups is not None and getattr(h, "spectral", False):
            # one power iteration per pass, in pass order (the reference's separate calls)
            states = [h.spectral_state() for _ in groups]
            sigma, u, v = tuple(s[0] for s in states), tuple(s[1] for s in states), tuple(s[2] for s in states)
        else:
            sigma, u, v = h.spectral_state()
        w = h.master()
        gamma = self.bn.weight if self.bn is not None else None
        beta = self.bn.bias if self.bn is not None else None
        y = F.LayerFn.apply(x, w, h.bias, gamma, beta, sigma, u, v, self, groups, cond)
        if self.kind == "dense":
            if self.out_mode == "f32":
                y = F.UnpadFn.apply(y, 0, self.cout) if (y.shape[1] != self.cout or y.dtype != torch.float32) else y
        return y


def prepack_dense(layers, parts=("fwd", "bwd")):
    """The stale operand copies of the small fp32 dense layers in `layers`, ALL in one launch (cpcsv_pack_dense_many) instead of one or two
    launches per layer at each layer's first use after the optimiser step. Same bytes as KernelLayer.packs() writes; the
    capture bookkeeping (PACK_LOG / USE_LOG) sees the same entries."""
    jobs, done = [], []
    for lay in layers:
        w = lay.holder.master()
        if w is None or not w.is_cuda:
            continue
        ent = lay._dense_job(w, parts)
        if ent is not None:
            jobs.append(ent[0])
            done.append((lay, w, ent[1], ent[2]))
    if not jobs:
        return
    with torch.no_grad():
        K.pack_dense_many(jobs)
    for lay, w, stale, key in done:
        if USE_LOG is not None:
            USE_LOG.append((lay, w, L.F32, stale))
        if PACK_LOG is not None:
            PACK_LOG.append((lay, w, L.F32, stale))
        lay._packs = lay._pack_bufs[(L.F32, w.device)]
        for part in stale:
            lay._key[part] = key


def _layer_for(holder, bn, act, up, head=False, out_mode=None, in_hw=None):
    """Create (once) the KernelLayer for a holder in a given fusion context."""
    key = (id(bn), act, up, head, out_mode, in_hw)
    lay = holder._layers.get(key)
    if lay is not None:
        return lay
    if isinstance(holder, Linear):
        lay = KernelLayer(holder, bn, act, 0, "dense", holder.cin, holder.cout, 1, 1, None, None,
                          out_mode or "f32", "Linear(%d->%d)" % (holder.cin, holder.cout))
    else:
        geom = F.ConvGeom(holder.k, holder.stride, holder.pad, up)
        if head:
            lay = _HeadConv(holder, bn, act)
        else:
            ntaps = geom.kh * geom.kw
            lay = KernelLayer(holder, bn, act, up, "conv", holder.cin, holder.cout, ntaps, ntaps, None, geom, out_mode or "T",
                              "Conv(%d->%d,k%s,s%s)" % (holder.cin, holder.cout, holder.k, holder.stride))
    holder._layers[key] = lay
    return lay


class _HeadConv:
    """A conv whose window covers the whole (padded) input so the output is 1x1 (cate_classify
    model.py:520, outlogits.3 model.py:79): run as a Linear over the flattened NHWC map."""

    def __init__(self, holder, bn, act):
        self.holder, self.bn, self.act = holder, bn, act
        self._by_hw = {}

    def __call__(self, x):
        h = self.holder
        if x.stride(1) == 1 and x.shape[1] in (h.cin, pad8(h.cin)):   # NCHW-shaped view of NHWC storage
            x = x.permute(0, 2, 3, 1)
        n, ih, iw, _ = x.shape
        lay = self._by_hw.get((ih, iw))
        if lay is None:
            geom = F.ConvGeom(h.k, h.stride, h.pad, 0)
            if geom.out_hw(ih, iw) != (1, 1):
                raise RuntimeError("head conv expects a 1x1 output, got %s" % (geom.out_hw(ih, iw),))
            tapmap = [((y + h.pad) * h.k + (x_ + h.pad)) if (y + h.pad < h.k and x_ + h.pad < h.k) else -1
                      for y in range(ih) for x_ in range(iw)]
            lay = KernelLayer(h, self.bn, self.act, 0, "dense", h.cin, h.cout, h.k * h.k, ih * iw, tapmap, None,
                              "f32", "HeadConv(%d->%d)" % (h.cin, h.cout))
            # one output, window == map, sigmoid, no BatchNorm: the critics' logit layer (csrc/head.hip)
            lay.logit_head = bool(_LOGIT_HEAD and h.cout == 1 and self.bn is None and self.act == L.ACT_SIGMOID
                                  and tapmap == list(range(ih * iw)) and h.k * h.k == ih * iw)
            self._by_hw[(ih, iw)] = lay
        if lay.logit_head and x.is_cuda:
            groups = current_groups()
            if groups is not None and len(groups) < 2:
                groups = None
            if groups is not None and getattr(h, "spectral", False):
                states = [h.spectral_state() for _ in groups]        # one power iteration per reference call, in call order
                sigma, u, v = tuple(s[0] for s in states), tuple(s[1] for s in states), tuple(s[2] for s in states)
            else:
                sigma, u, v = h.spectral_state()
            y = F.LogitHeadFn.apply(x.contiguous().view(n, -1), h.master(), h.bias, sigma, u, v, lay, groups)
            return y.view(n, h.cout, 1, 1)
        y = lay(x)                                  # fp32 [N, Cout]
        return y.view(n, h.cout, 1, 1)


# ------------------------------------------------------------------------------------------------
# fused Sequential
# ------------------------------------------------------------------------------------------------
class FusedSequential(nn.Sequential):
    """nn.Sequential with the reference's child indices; forward executes fused LayerFn nodes.

    `out_mode` of the LAST layer: 'T' keeps the padded compute-dtype tensor (feeds another fused
    layer), 'f32' returns an unpadded fp32 matrix (dense chains feeding fp32 glue)."""

    def __init__(self, *mods, out_mode=None, head_last=False):
        super().__init__(*mods)
        self._out_mode = out_mode
        self._head_last = head_last

    def _plan(self):
        plan = self.__dict__.get("_cached_plan")
        if plan is not None:
            return plan
        kids = list(self.children())
        plan, i, up = [], 0, 0
        while i < len(kids):
            m = kids[i]
            if isinstance(m, Upsample):
                up, i = 1, i + 1
                continue
            if not isinstance(m, (Conv2d, Linear)):
                raise RuntimeError("FusedSequential: unexpected %s at %d" % (type(m).__name__, i))
            j, bn, act = i + 1, None, L.ACT_NONE
            if j < len(kids) and isinstance(kids[j], _BatchNorm):
                bn, j = kids[j], j + 1
            if j < len(kids) and type(kids[j]) in _ACT_OF:
                act, j = _ACT_OF[type(kids[j])], j + 1
            last = j >= len(kids)
            head = last and self._head_last
            out_mode = self._out_mode if last else None
            plan.append(_layer_for(m, bn, act, up, head=head, out_mode=out_mode))
            up, i = 0, j
        self.__dict__["_cached_plan"] = plan
        return plan

    def forward(self, x):
        for lay in self._plan():
            x = lay(x)
        return x


def dense_input(*pieces, dtype=None):
    """fp32 [B,k_i] pieces -> one padded matrix in `dtype` (default: the compute dtype), one launch."""
    return F.PadCastFn.apply(dtype or tdtype(), *pieces)


class GRUCell(nn.Module):
    """nn.GRUCell (model.py:223-224): two dense GEMMs + the gate kernel. Default init U(+-1/sqrt(H))
    (the reference's weights_init does not touch it)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        self.weight_ih = nn.Parameter(torch.empty(3 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh = nn.Parameter(torch.empty(3 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))
        self.bias_hh = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))
        self._lay = None

    def _layers(self):
        if self._lay is None:
            class _H:   # minimal holder protocol for KernelLayer
                def __init__(s, w, b, cin, cout):
                    s.w, s.bias, s.cin, s.cout = w, b, cin, cout

                def master(s):
                    return s.w

                def spectral_state(s):
                    return None, None, None
            hi = _H(self.weight_ih, self.bias_ih, self.input_size, 3 * self.hidden_size)
            hh = _H(self.weight_hh, self.bias_hh, self.hidden_size, 3 * self.hidden_size)
            mk = lambda h, nm: KernelLayer(h, None, L.ACT_NONE, 0, "dense", h.cin, h.cout, 1, 1, None, None, "f32pad", nm)
            self._lay = (mk(hi, "GRU.ih"), mk(hh, "GRU.hh"), hi, hh)
        li, lh, hi, hh = self._lay
        hi.w, hi.bias, hh.w, hh.bias = self.weight_ih, self.bias_ih, self.weight_hh, self.bias_hh
        return li, lh

    def in_dtype(self):
        return self._layers()[0].in_dtype()

    def input_gates(self, x):
        """W_ih x + b_ih for any number of rows: the inputs of all time steps of a sequence are known up front, so
        the caller runs this ONCE on the time-major stack (one GEMM forward, one weight gradient backward)."""
        return self._layers()[0](x)

    def step(self, gi, h):
        """One recurrence step from precomputed input gates."""
        return F.GruPointFn.apply(gi, self._layers()[1](h), h, self.hidden_size)

    def sequence(self, gi_all, h0):
        """All steps of a recurrence from their input gates [T,B,ldg] and the initial state [B,ldh] (padded layout) -> the states
        [T,B,ldh]: one launch per step forward, two per step + one for the weight gradients backward (cpcsv.functional.GruSeqFn).
        Falls back to step() for more than 64 rows or a non-fp32 W_hh layer."""
        lh = self._layers()[1]
        t_, b = gi_all.shape[0], gi_all.shape[1]
        if not (_GRU_SEQ and lh.compute_f32 and b <= 64 and gi_all.is_cuda and h0.shape[1] == lh.cin_s):
            hs, h = [], h0
            for t in range(t_):
                h = self.step(gi_all[t], h)
                hs.append(h)
            return torch.stack(hs, 0)
        return F.GruSeqFn.apply(gi_all, h0, self.weight_hh, self.bias_hh, lh, self.hidden_size)

    def forward(self, x, h):
        return self.step(self.input_gates(x), h)
