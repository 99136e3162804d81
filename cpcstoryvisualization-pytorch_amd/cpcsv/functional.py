"""autograd.Functions of the HIP path. Every forward/backward here is a sequence of C-ABI
kernel launches (cpcsv.kernels); torch supplies memory, the autograd graph and the stream.

Activation tensors are NHWC `[N, H, W, Cs]` (Cs = channels padded to 8, pads zero) in the compute
dtype; dense activations are `[B, Ks]`. Layer semantics follow the reference call sites cited in
cpcsv/modules.py and model.py.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import kernels as K
from . import runtime as _runtime
from .runtime import (branch_id, branch_role, dcode, defer_late, flush_late, forced_stream, fork_to, keep_alive, late_stream,
                      note_side_write, pad8, require_gpu, tdtype, wait_side_writes, wgrad_stream)


_POISON = os.environ.get("CPCSV_POISON", "0") == "1"


def _empty(shape, dtype, dev, zero=False):
    if zero:
        return torch.zeros(shape, dtype=dtype, device=dev)
    t = torch.empty(shape, dtype=dtype, device=dev)
    return t.fill_(float("nan")) if _POISON and t.is_floating_point() else t


def _empty_like(t):
    """torch.empty_like; CPCSV_POISON=1 fills with NaN so a read of a never-written element surfaces in the tests."""
    r = torch.empty_like(t)
    return r.fill_(float("nan")) if _POISON and r.is_floating_point() else r


# ------------------------------------------------------------------------------------------------
# geometry of a convolution expressed as gather-GEMM launches
# ------------------------------------------------------------------------------------------------
class ConvGeom:
    """Kernel / stride / padding of a 2-D convolution, square (ints) or not (pairs (h, w): the order critic's temporal
    convs run as (3,1)-kernel, (2,1)-stride convs over [story][T][H*W] "images")."""

    def __init__(self, k, stride, pad, up=0):
        pair = lambda v: (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))
        (self.kh, self.kw), (self.sh, self.sw), (self.ph, self.pw), self.up = pair(k), pair(stride), pair(pad), up
        # square views (sub-pixel / thin-kernel eligibility tests); -1 when not square
        self.k = self.kh if self.kh == self.kw else -1
        self.s = self.sh if self.sh == self.sw else -1
        self.p = self.ph if self.ph == self.pw else -1

    def out_hw(self, ih, iw):
        eh, ew = ih << self.up, iw << self.up
        return (eh + 2 * self.ph - self.kh) // self.sh + 1, (ew + 2 * self.pw - self.kw) // self.sw + 1

    def fwd_taps(self):
        taps = [(u - self.ph, v - self.pw, u * self.kw + v) for u in range(self.kh) for v in range(self.kw)]
        if _PATCH_S2 and (self.kh, self.kw, self.sh, self.sw, self.ph, self.pw) == (4, 4, 2, 2, 1, 1):
            # 4x4 stride-2 pad-1 (the critics' towers): taps grouped by the PARITY of the input pixel they read, four runs of
            # four - the order the patch-resident main loop wants (csrc/gemm.hip conv_patch_kernel: one LDS-resident input
            # patch per parity class serves its 4 taps); any order is the same convolution (the weight slice rides in the tap)
            taps.sort(key=lambda t: (((t[0] + 1) & 1), ((t[1] + 1) & 1)))
        return taps

    def dgrad_launches(self, ih, iw):
        """[(taps, MH, MW, pool, scatter)] producing dX (stored IHxIW) from dY."""
        kh, kw, ph, pw, sh, sw = self.kh, self.kw, self.ph, self.pw, self.sh, self.sw
        if sh == 1 and sw == 1:
            taps = [(ph - u, pw - v, u * kw + v) for u in range(kh) for v in range(kw)]
            return [(taps, ih << self.up, iw << self.up, self.up, None)]
        out = []
        for py in range(sh):
            for px in range(sw):
                taps = [((py + ph - u) // sh, (px + pw - v) // sw, u * kw + v)
                        for u in range(kh) for v in range(kw)
                        if (py + ph - u) % sh == 0 and (px + pw - v) % sw == 0]
                mh, mw = (ih - py + sh - 1) // sh, (iw - px + sw - 1) // sw
                if taps and mh > 0 and mw > 0:
                    out.append((taps, mh, mw, 0, (ih, iw, sh, sw, py, px)))
        return out

    def dgrad_covers_all(self):
        if self.sh == 1 and self.sw == 1:
            return True
        rows = all(any((py + self.ph - u) % self.sh == 0 for u in range(self.kh)) for py in range(self.sh))
        cols = all(any((px + self.pw - v) % self.sw == 0 for v in range(self.kw)) for px in range(self.sw))
        return rows and cols


# ---- sub-pixel form of nearest-x2 upsample + conv3x3 (model.py:26-34) ----------------------------------------
# Output pixel (2i+a, 2j+b) only ever sees the 2x2 low-res neighbourhood (i+a+p-1, j+b+q-1), p,q in {0,1}: the 3x3
# taps that land on the same low-res pixel are pre-summed. 4 parity phases x 4 taps = 16 MACs per output pixel
# instead of 36 for 4 outputs x 9 taps: 2.25x fewer FLOPs in forward, dgrad and wgrad, identical result up to
# fp32 summation order. Slice index s = ((a*2+b)*2+p)*2+q.
_SUB_U = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
_SUB_SLICES = [(a, b, p, q) for a in (0, 1) for b in (0, 1) for p in (0, 1) for q in (0, 1)]
SUB_MASKS = [sum(1 << (u * 3 + v) for u in _SUB_U[(a, p)] for v in _SUB_U[(b, q)]) for a, b, p, q in _SUB_SLICES]
SUB_FWD_TAPS = [(a + p - 1, b + q - 1, s) for s, (a, b, p, q) in enumerate(_SUB_SLICES)]
SUB_PHASES = [(4 * (a * 2 + b), 4, a, b) for a in (0, 1) for b in (0, 1)]
SUB_DGRAD_TAPS = [(2 - a - 2 * p, 2 - b - 2 * q, s) for s, (a, b, p, q) in enumerate(_SUB_SLICES)]
SUB_WGRAD_TAPS = [(a + p - 1, b + q - 1, s, a | (b << 4)) for s, (a, b, p, q) in enumerate(_SUB_SLICES)]


_WG_BLOCKS = 512          # (256: +0.33 ms, 1024: +0.60 ms per step at round 6's HEAD, profiles/r06_knob_sweep.txt; env override retired)
_WG_MINROWS = 512


def _splits_for(tiles, m):
    # ~2 blocks per CU in total; every pixel slice should still loop >= 8 K tiles (64 pixels each) so that the
    # prologue, the epilogue and the fp32 atomics of the extra slices stay a small part of the block
    return int(max(1, min(_WG_BLOCKS // max(tiles, 1), m // _WG_MINROWS)))


# CPCSV_PATCH (csrc/gemm.hip): 0 = streaming gather-GEMM everywhere, 1 (default) = patch-resident main loop for the stride-1
# phase launches with more than 64 output columns, 2 = also for 4x4 stride-2 windows, whose taps then travel in parity-class order
_PATCH_S2 = int(os.environ.get("CPCSV_PATCH", "2")) >= 2
_THIN = os.environ.get("CPCSV_THIN", "1") != "0"
_ROWS_INLINE = os.environ.get("CPCSV_ROWS_INLINE_WG", "1") != "0"
_DENSE_ROWS = os.environ.get("CPCSV_DENSE_ROWS", "1") != "0"      # fp32 dense layers over <= 64 rows: one cpcsv_dense_rows launch
_THIN4_DGRAD = os.environ.get("CPCSV_THIN4_DGRAD", "1") != "0"
_THIN4_WGRAD = os.environ.get("CPCSV_THIN4_WGRAD", "1") != "0"      # A/B switch of the critics' first-conv weight-gradient kernel
_PAIR = os.environ.get("CPCSV_WGRAD_PAIR", "1") != "0"
_EARLY_BWD_PACK = os.environ.get("CPCSV_EARLY_BWD_PACK", "1") != "0"
_BN_FOLD = os.environ.get("CPCSV_BN_FOLD", "1") != "0"         # few partial rows: bn_apply sums them itself (no finalize launch)
_BN_FOLD_ROWS = 16         # (32 / 64 folded rows measured slower in round 4, 8 / 32 neutral in round 6; env override retired)


def flush_stash(mod):
    """A first pass parked for a shared weight-gradient launch whose partner never came (or came with another shape):
    run it on its own."""
    st = mod.fused_stash
    if st is not None:
        dz1, x1, wd, g = st
        mod.fused_stash = None
        K.wgrad_run(wd, dz1, x1, g, accumulate=0)


def _thin_kind(mod, x, has_bn, bias, sigma):
    """0 = general gather-GEMM; 1 = streaming 3x3 conv with <= 4 output channels (StoryGAN.img / img_seg); 2 = the
    critics' first 4x4 stride-2 conv over an 8-stored-channel image. bf16 only (fp32 parity mode keeps the exact-f32
    MFMA GEMM), no BatchNorm, no bias; kind 1 also has no spectral norm."""
    if not _THIN or x.dtype != torch.bfloat16 or has_bn or bias is not None or mod.subpixel or mod.out_f32:
        return 0
    g = mod.geom
    key = ("thin", tuple(x.shape))
    kind = mod.descs.get(key)
    if kind is None:
        n, ih, iw, cs = x.shape
        kind = 0
        if g.k == 3 and g.s == 1 and g.p == 1 and g.up == 0 and sigma is None and K.thin_supported(0, cs, mod.cout, ih, iw):
            kind = 1
        elif g.k == 4 and g.s == 2 and g.p == 1 and g.up == 0 and K.thin_supported(1, cs, mod.cout, ih, iw):
            kind = 2
        mod.descs[key] = kind
    return kind


def _thin4_slabs(mod, xshape, cout, dev):
    """Per-block partial-sum workspace of the streaming weight gradient of the critics' first conv (None: shape not served)."""
    key = ("thin4_slabs", tuple(xshape))
    if key not in mod.descs:
        n, ih, iw, _ = xshape
        ns = K.thin4x4s2_wgrad_slabs(n, ih, iw) if _THIN4_WGRAD else 0
        mod.descs[key] = torch.empty(ns * cout * 128, dtype=torch.float32, device=dev) if ns > 0 else None
    return mod.descs[key]


# ------------------------------------------------------------------------------------------------
# conv / linear (+ spectral norm scale, + bias, + BatchNorm(train), + activation) as ONE autograd node
# ------------------------------------------------------------------------------------------------
def _cum(counts, unit):
    out, t = [0], 0
    for c in counts:
        t += c * unit
        out.append(t)
    return out


def _as_list(x, n):
    """per-group view of an argument that is one tensor (single pass) or a tuple of n tensors (row groups)"""
    if x is None:
        return None
    return list(x) if isinstance(x, (tuple, list)) else [x] * n


class LayerFn(Function):
    """y = act(BN(conv(x, W) * (1/sigma) + b)).  `mod` is a cpcsv.modules.KernelLayer.

    `groups` (tuple of leading-dimension counts, or None): x holds several passes of the layer back to back - the real and
    the fake batch of a critic, the story and the image half of a generator pass. ONE set of launches serves them all;
    BatchNorm statistics, running-statistics updates (in group order) and the spectral-norm scale are per pass, exactly as
    the reference's separate calls. sigma / u / v are then tuples with one entry per pass (call order)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, sigma, u, v, mod, groups=None, cond=None):
        require_gpu(x)
        x = x.contiguous()
        dev, T = x.device, x.dtype
        dt = dcode(x)
        ctx.cond = None
        if cond is not None:
            return LayerFn._cond_forward(ctx, x, weight, bias, gamma, beta, sigma, u, v, mod, groups, cond)
        # small fp32 dense layers (text / motion encoders, GRU products) of a differentiable pass: rebuild the data-gradient operand
        # copy NOW, where the forward hides behind other work, instead of at the tail of the backward pass, whose chain of tiny
        # launches is the critical path there (three ~15 us transposing packs per generator backward)
        early_bwd = _EARLY_BWD_PACK and mod.compute_f32 and ctx.needs_input_grad[0]       # (a backward pass will want dX)
        fwd, _, _ = mod.packs(weight, dt, "both" if early_bwd else "fwd")
        conv = mod.kind == "conv"
        cout, cout_s = mod.cout, pad8(mod.cout)
        sub = conv and mod.subpixel
        groups = tuple(int(g) for g in groups) if groups is not None and len(groups) > 1 else None
        if groups is not None and sum(groups) != x.shape[0]:
            raise RuntimeError("%s: row groups %s do not add up to %d rows" % (mod.name, groups, x.shape[0]))
        counts = groups if groups is not None else (x.shape[0],)
        ng = len(counts)
        if conv:
            n, ih, iw, cs = x.shape
            oh, ow = mod.geom.out_hw(ih, iw)
            m = n * oh * ow
            oshape = (n, oh, ow, cout_s)
            in_unit, out_unit = (ih * iw if sub else oh * ow), oh * ow
            if sub:     # 4 parity phases over the LOW-res grid, each a 2x2 conv writing its quarter of the output
                taps, geo = SUB_FWD_TAPS, dict(MH=ih, MW=iw, IH=ih, IW=iw, scatter=(oh, ow, 2, 2, 0, 0), phases=SUB_PHASES)
            else:
                taps, geo = mod.geom.fwd_taps(), dict(MH=oh, MW=ow, IH=ih, IW=iw, sy=mod.geom.sh, sx=mod.geom.sw, up=mod.geom.up)
        else:
            m, cs = x.shape
            oshape = (m, cout_s)
            in_unit = out_unit = 1
            taps, geo = [(0, 0, 0)], {}
        if cs != mod.k_stored:
            raise RuntimeError("%s: input has %d stored channels, layer expects %d" % (mod.name, cs, mod.k_stored))
        if getattr(mod, "dgrad_cols", None) and ctx.needs_input_grad[0] and getattr(x, "_cpcsv_live_cols", None) != mod.dgrad_cols:
            # the data gradient of this layer is computed for its first `dgrad_cols` input columns only (the rest of dX stays
            # unwritten): legal only behind a producer that reads exactly those columns and says so (CondConcatFn / CondTripletFn)
            raise RuntimeError("%s computes dX for its first %d columns only, but its input does not come from a producer that "
                               "declares it reads just those (got %r)" % (mod.name, mod.dgrad_cols, getattr(x, "_cpcsv_live_cols", None)))
        raw_f32 = mod.out_f32 and T != torch.float32
        rdtype = torch.float32 if mod.out_f32 else T
        has_bn = gamma is not None
        y_raw = _empty(oshape, rdtype, dev)       # channel pads are written (as zeros) by the GEMM epilogue
        sig = _as_list(sigma, ng)                 # per-pass {sigma, 1/sigma}
        thin = _thin_kind(mod, x, has_bn, bias, sigma) if conv else 0
        ctx.groups, ctx.counts, ctx.sig, ctx.us, ctx.vs = groups, counts, sig, _as_list(u, ng), _as_list(v, ng)
        ctx.in_unit, ctx.out_unit = in_unit, out_unit
        if thin:
            # HBM-bound layers with a degenerate GEMM dimension (csrc/thin.hip): the input crosses HBM -> LDS once
            if thin == 1:
                K.thin3x3_fwd(x, fwd, y_raw, n, ih, iw, cs, cout, mod.act)
            elif sig is None or ng == 1:
                K.thin4x4s2_fwd(x, fwd, y_raw, sig[0][1:] if sig is not None else None, n, ih, iw, cout, mod.act)
            else:                                  # per-pass 1/sigma: one launch per pass on its rows
                r0 = 0
                for g in range(ng):
                    K.thin4x4s2_fwd(x[r0:r0 + counts[g]], fwd, y_raw[r0:r0 + counts[g]], sig[g][1:], counts[g], ih, iw, cout, mod.act)
                    r0 += counts[g]
            ctx.mod, ctx.has_bn, ctx.conv, ctx.m, ctx.sub, ctx.branch, ctx.thin = mod, False, True, m, False, branch_id(), thin
            ctx.xshape = tuple(x.shape)
            ctx.save_for_backward(x, weight, bias, gamma, beta, None, y_raw, None)
            return y_raw
        if _DENSE_ROWS and not conv and dt == L.F32 and m <= 64 and groups is None and sig is None and not raw_f32:
            # a handful of rows in exact fp32 (text / motion encoders, GRU recurrences): one launch (cpcsv_dense_rows) instead
            # of the split-K GEMM + slab pass; BatchNorm partials per block of 16 rows
            stats, tiles, nph, desc, bg_out = None, None, 1, None, None
            if has_bn:
                pstride = (4 + 2 * L.BN_SUM_COPIES) * cout_s
                bg_out = K.bn_groups(_cum(counts, out_unit), pstride)
                if mod.bn.training:
                    tiles = _cum([(c + 15) // 16 for c in counts], 1)
                    mtiles = tiles[-1]
                    stats = _empty((mtiles, 2, cout_s), torch.float32, dev)
            K.dense_rows(x, fwd, y_raw, m, cout, cs, None, bias, 0 if has_bn else mod.act, stats, cout_s)
            return LayerFn._finish_forward(ctx, mod, x, weight, bias, gamma, beta, y_raw, has_bn, conv, sub, m, ng, counts, in_unit, out_unit,
                                           cout, cout_s, dev, desc, stats, tiles, nph, bg_out, mtiles if stats is not None else 0)
        key = ("fwd", tuple(x.shape), dt, has_bn, branch_id(), groups)
        desc = mod.descs.get(key)
        if desc is None:
            desc = mod.descs[key] = K.gemm_desc(None, None, None, dtype=dt, M=(m // 4 if sub else m), N=cout, Cs=cs,
                                                ldb=fwd.shape[1], ldc=cout_s, taps=taps, act=(0 if has_bn else mod.act),
                                                out_f32=int(raw_f32), **geo)
            desc._algo = 2.25 if sub else 1.0          # reference-algorithm FLOPs / executed FLOPs (bench metering)
            if groups is not None:
                K.set_row_groups(desc, _cum(counts, in_unit))
        K.bind(desc, x, fwd, y_raw, sig[0][1:] if sig is not None else None, bias)
        if groups is not None:
            K.bind_group_alpha(desc, [sg[1:] for sg in sig] if sig is not None else None)
        stats = None
        ws = K.gemm_nt_auto(desc, m, dev)        # split-K plan (+ fp32 workspace) for few-tile / long-K shapes
        bg_out = None
        if has_bn:
            pstride = (4 + 2 * L.BN_SUM_COPIES) * cout_s
            bg_out = K.bn_groups(_cum(counts, out_unit), pstride)       # output rows of the passes
        if has_bn and mod.bn.training:
            mt = K.gemm_mtile(desc)
            if sub and desc.splitk <= 1:          # one partial per (phase, M tile of the low-resolution grid)
                tiles, nph = _cum([(c * in_unit + mt - 1) // mt for c in counts], 1), 4
            elif desc.splitk > 1:                 # the split-K epilogue pass: partials per block of OUTPUT rows
                tiles, nph = _cum([(c * out_unit + mt - 1) // mt for c in counts], 1), 1
            else:
                tiles, nph = _cum([(c * in_unit + mt - 1) // mt for c in counts], 1), 1
            mtiles = nph * tiles[-1]
            stats = _empty((mtiles, 2, cout_s), torch.float32, dev)
            desc.stats, desc.ldstat = stats.data_ptr(), cout_s
        K.gemm_nt(desc)
        del ws
        return LayerFn._finish_forward(ctx, mod, x, weight, bias, gamma, beta, y_raw, has_bn, conv, sub, m, ng, counts, in_unit, out_unit,
                                       cout, cout_s, dev, desc, stats, tiles if stats is not None else None, nph if stats is not None else 1,
                                       bg_out, mtiles if stats is not None else 0)

    @staticmethod
    def _cond_forward(ctx, x, weight, bias, gamma, beta, sigma, u, v, mod, groups, cond):
        """D_GET_LOGITS' 3x3 conv (reference model.py:75-80,89-92) in factored form - csrc/condhead.hip, include/cpcsv_hip.h
        cpcsv_cond_head. x = the DISTINCT feature maps [S,4,4,Cf] ([real | fake] of a critic update, or one batch), cond = the
        distinct condition rows [Nc,E] fp32 (detached by every caller: no gradient); `groups` = (N, N-1, N): the real / wrong /
        fake calls of miscc/utils.py:70-84 (wrong = real features [0, N-1) with conditions [1, N)), or None: one call.
        Launches: condition rows -> operand dtype, their 9 per-tap products (one GEMM, bcol_rows), the feature conv over the
        distinct maps (K = 9 Cf instead of 9 (Cf + E); fp32 K-slice slabs), and cpcsv_cond_head_fwd: assembly of every call's
        conv output + train-mode BatchNorm + activation."""
        dev, T, dt = x.device, x.dtype, dcode(x)
        s_, ih, iw, cf = x.shape
        cout, cout_s, cin_s = mod.cout, pad8(mod.cout), mod.cin_s
        e_s = cin_s - cf
        fwd, _, _ = mod.packs(weight, dt, "fwd")
        if groups is not None:
            n = int(groups[0])
            counts, feat0, cond0 = (n, n - 1, n), (0, 0, n), (0, 1, 0)
        else:
            counts, feat0, cond0 = (s_,), (0,), (0,)
        ng, nc = len(counts), cond.shape[0]
        total = sum(counts)
        m = total * ih * iw
        sig = _as_list(sigma, ng)
        ctx.groups, ctx.counts, ctx.sig, ctx.us, ctx.vs = (counts if ng > 1 else None), counts, sig, _as_list(u, ng), _as_list(v, ng)
        ctx.in_unit = ctx.out_unit = ih * iw
        ctx.cond_geo = (counts, feat0, cond0, s_, nc, cf, e_s)
        # the condition rows in the operand dtype, zero channel pads
        cond_t = _empty((nc, e_s), T, dev)
        K.concat_pad([cond.contiguous().float()], cond_t, nc, e_s)
        # per-tap condition products pt[c][tap][o] = <cond[c], W[o][tap][Cf:]>
        key = ("cond_pt", nc, dt)
        dp = mod.descs.get(key)
        if dp is None:
            dp = mod.descs[key] = K.gemm_desc(None, None, None, dtype=dt, M=nc, N=9 * cout, Cs=e_s, ldb=fwd.shape[1], ldc=9 * cout,
                                              taps=[(0, 0, 0)], out_f32=1)
            dp.bcol_rows, dp.bcol_koff, dp.patch = cout, cin_s, -1
            dp._algo = 0.0                      # (bench metering: the reference algorithm's FLOPs of this layer are booked on the feature conv)
            dp._pt = torch.empty((nc, 9, cout), dtype=torch.float32, device=dev)
        pt = dp._pt
        K.bind(dp, cond_t, fwd, pt)
        dp.B = fwd.data_ptr() + cf * fwd.element_size()
        K.gemm_nt(dp)
        # the feature conv over the distinct maps
        key = ("cond_feat", s_, dt)
        df = mod.descs.get(key)
        if df is None:
            df = mod.descs[key] = K.gemm_desc(None, None, None, dtype=dt, M=s_ * ih * iw, N=cout, Cs=cf, ldb=fwd.shape[1], ldc=cout_s,
                                              taps=mod.geom.fwd_taps(), MH=ih, MW=iw, IH=ih, IW=iw, out_f32=1)
            df.wstride, df.patch = cin_s, -1
            df._algo = float(m * cin_s) / float(s_ * ih * iw * cf)
            sk = K.plan_splitk(df, 64 if dt == L.BF16 else 32)
            df._ws = torch.empty((max(sk, 1), s_ * ih * iw, cout_s), dtype=torch.float32, device=dev)
            if sk > 1:
                df.splitk, df.ws, df.ldws, df.ws_rows, df.slabs_only = sk, df._ws.data_ptr(), cout_s, s_ * ih * iw, 1
            df._sk = sk
        ws = df._ws
        K.bind(df, x, fwd, ws)
        K.gemm_nt(df)
        # assembly + BatchNorm + activation
        key = ("cond_head", counts)
        dh = mod.descs.get(key)
        if dh is None:
            dh = mod.descs[key] = K.cond_head_desc(counts, feat0, cond0)
        pstride = (4 + 2 * L.BN_SUM_COPIES) * cout_s
        bnbuf = _empty((ng, 4 + 2 * L.BN_SUM_COPIES, cout_s), torch.float32, dev)
        y_raw = _empty((total, ih, iw, cout_s), T, dev)
        y = _empty_like(y_raw)
        K.cond_head_fwd(dh, ws, max(df._sk, 1), pt, [sg[1:] for sg in sig] if sig is not None else None, y_raw, y, gamma, beta,
                        mod.bn.running_mean, mod.bn.running_var, bnbuf, pstride, any(ctx.needs_input_grad), cout, mod.act, mod.bn.eps,
                        mod.bn.momentum)
        for _ in range(ng):
            mod.bn.note_batch()
        ctx.cond = cond_t
        ctx.mod, ctx.has_bn, ctx.conv, ctx.m, ctx.sub, ctx.branch, ctx.thin = mod, True, True, m, False, branch_id(), 0
        ctx.xshape = tuple(x.shape)
        ctx.save_for_backward(x, weight, bias, gamma, beta, y_raw, None, bnbuf)
        return y

    @staticmethod
    def _finish_forward(ctx, mod, x, weight, bias, gamma, beta, y_raw, has_bn, conv, sub, m, ng, counts, in_unit, out_unit, cout, cout_s,
                        dev, desc, stats, tiles, nph, bg_out, mtiles):
        """BatchNorm finalize + apply behind the product launch, bookkeeping for backward (shared by the GEMM and dense_rows paths)."""
        pstride = (4 + 2 * L.BN_SUM_COPIES) * cout_s
        y = y_raw
        bnbuf = None
        if has_bn:
            # per pass: rows mean, invstd, scale, shift, then the [COPIES][2][Cs] accumulators of the backward pass (zeroed by finalize)
            bnbuf = _empty((ng, 4 + 2 * L.BN_SUM_COPIES, cout_s), torch.float32, dev)
            if mod.bn.training and _BN_FOLD and stats is not None and tiles is not None and tiles[-1] * nph <= _BN_FOLD_ROWS and branch_role() is None:
                # a handful of statistics partials (dense layers, 4x4 maps): bn_apply sums them itself - no bn_finalize launch
                # (fixed summation order: also in the deterministic mode)
                bg = K.bn_groups(_cum(counts, out_unit), pstride, tiles=tiles, nph=nph)
                y = _empty_like(y_raw)
                K.bn_apply_partials(y_raw, y, stats, cout_s, gamma, beta, mod.bn.running_mean, mod.bn.running_var, bnbuf,
                                    bnbuf[0, 4:] if any(ctx.needs_input_grad) else None, m, cout, cout_s, mod.act, mod.bn.eps,
                                    mod.bn.momentum, bg)
                for _ in range(ng):
                    mod.bn.note_batch()
                ctx.mod, ctx.has_bn, ctx.conv, ctx.m, ctx.sub, ctx.branch, ctx.thin = mod, has_bn, conv, m, sub, branch_id(), 0
                ctx.xshape = tuple(x.shape)
                ctx.save_for_backward(x, weight, bias, gamma, beta, y_raw, None, bnbuf)
                return y
            if mod.bn.training:
                role = branch_role()
                if role == "second" and getattr(mod.bn, "_order_ev", None) is not None:
                    torch.cuda.current_stream().wait_event(mod.bn._order_ev)      # running stats: first half, then this one
                bg_fin = K.bn_groups(_cum(counts, in_unit if (sub and desc.splitk <= 1) else out_unit), pstride, tiles=tiles, nph=nph)
                K.bn_finalize(stats, mtiles, cout_s, m, gamma, beta, mod.bn.running_mean, mod.bn.running_var,
                              bnbuf[0, 0], bnbuf[0, 1], bnbuf[0, 2], bnbuf[0, 3], cout, cout_s, mod.bn.eps, mod.bn.momentum, True,
                              bwd_sums=bnbuf[0, 4:] if any(ctx.needs_input_grad) else None, groups=bg_fin)
                if role == "first":
                    mod.bn._order_ev = torch.cuda.Event()
                    mod.bn._order_ev.record()
                for _ in range(ng):
                    mod.bn.note_batch()
            else:   # eval: running statistics (tiny host-side vectors; not on the training path)
                inv = torch.rsqrt(mod.bn.running_var + mod.bn.eps)
                bnbuf.zero_()
                bnbuf[:, 0, :cout] = mod.bn.running_mean
                bnbuf[:, 1, :cout] = inv
                bnbuf[:, 2, :cout] = gamma * inv
                bnbuf[:, 3, :cout] = beta - mod.bn.running_mean * gamma * inv
            y = _empty_like(y_raw)
            K.bn_apply(y_raw, y, bnbuf[0, 2], bnbuf[0, 3], m, cout, cout_s, mod.act, groups=bg_out)
        ctx.mod, ctx.has_bn, ctx.conv, ctx.m, ctx.sub, ctx.branch, ctx.thin = mod, has_bn, conv, m, sub, branch_id(), 0
        ctx.xshape = tuple(x.shape)
        # BN layers keep the raw conv output (z and the activation mask are recomputed from it); others keep y
        ctx.save_for_backward(x, weight, bias, gamma, beta, y_raw if has_bn else None, None if has_bn else y, bnbuf)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        mod = ctx.mod
        x, weight, bias, gamma, beta, y_raw, y, bnbuf = ctx.saved_tensors
        sig, us, vs, counts, groups = ctx.sig, ctx.us, ctx.vs, ctx.counts, ctx.groups
        ng = len(counts)
        dev, T = x.device, x.dtype
        dt = dcode(x)
        m, cout, cout_s = ctx.m, mod.cout, pad8(mod.cout)
        dy = dy.contiguous()
        dgamma = dbeta = dbias = dw = dx = gw_bn = None
        # parameters whose .grad is a persistent buffer owned by the trainer (one flat buffer per net) get their
        # gradient ACCUMULATED in place by the kernels; autograd then sees None (no per-tensor add launches, static
        # pointers for the multi-tensor Adam table and the gradient all-reduce)
        direct = lambda p: p is not None and getattr(p, "_cpcsv_direct", False) and p.grad is not None
        want_w = ctx.needs_input_grad[1]
        folded = False            # 1/sigma already multiplied into dz (BatchNorm layers): the GEMMs below need no per-pass scale
        # ---- through BN / activation: dz = dL/d(conv output incl. bias) ----
        if ctx.has_bn:
            pstride = (4 + 2 * L.BN_SUM_COPIES) * cout_s
            bg = K.bn_groups(_cum(counts, ctx.out_unit), pstride, sigmas=sig)
            folded = sig is not None
            if not mod.bn.training:
                for g in range(ng):
                    K.fill_zero(bnbuf[g, 4:])
            K.bn_bwd_reduce(dy, y_raw, bnbuf[0, 0], bnbuf[0, 1], gamma, beta, bnbuf[0, 4:], m, cout, cout_s, mod.act, groups=bg)
            dz = _empty_like(y_raw)
            # spectral-normed conv + train-mode BN: sum(G .* W) of every pass comes out of this launch in closed form (no dot kernel)
            gw_bn = None
            if sig is not None and mod.bn.training and want_w:
                slots = getattr(mod, "gw_slots", None)
                if slots is not None and mod.fused and mod.fused_seen * ng + ng <= slots.numel():
                    # data-parallel runs: persistent slots behind the net's flat gradient buffer - mean-reduced with it before the update
                    gw_bn = slots[mod.fused_seen * ng:mod.fused_seen * ng + ng]
                else:
                    gw_bn = _empty((ng,), torch.float32, dev)
            if direct(gamma) and direct(beta):
                K.bn_bwd_apply(dy, y_raw, dz, bnbuf[0, 0], bnbuf[0, 1], gamma, beta, bnbuf[0, 4:], gamma.grad, beta.grad, m, cout,
                               cout_s, mod.act, accumulate=1, gw_out=gw_bn, eps=mod.bn.eps, groups=bg)
            else:
                dgb = _empty((2, cout), torch.float32, dev)
                K.bn_bwd_apply(dy, y_raw, dz, bnbuf[0, 0], bnbuf[0, 1], gamma, beta, bnbuf[0, 4:], dgb[0], dgb[1], m, cout, cout_s, mod.act,
                               gw_out=gw_bn, eps=mod.bn.eps, groups=bg)
                dgamma, dbeta = dgb[0], dgb[1]
        elif mod.act != L.ACT_NONE:
            dz = _empty_like(y)
            K.act_bwd(dy, y, dz, mod.act)
        else:
            dz = dy
        if dz.dtype != T:   # fp32 head output in bf16 mode: operands of the GEMMs are bf16
            dzt = _empty(dz.shape, T, dev)
            K.copy2d(dz, cout_s, 0, dzt, cout_s, 0, m, cout_s)
        else:
            dzt = dz
        _, bwd, lin = mod.packs(weight, dt, "bwd")
        cond_t, dZt = ctx.cond, None
        if cond_t is not None:
            # factored head conv (_cond_forward): dY of the feature conv's GEMMs = the calls' dz summed per DISTINCT feature map,
            # dY of the condition columns' weight gradient = dz summed per distinct condition row, tap by tap over its live pixels
            c_counts, c_feat0, c_cond0, c_s, c_nc, c_cf, c_es = ctx.cond_geo
            if len(c_counts) == 1 and not want_w:
                dF = dzt
            else:
                dF = _empty((c_s,) + tuple(dzt.shape[1:]), T, dev)
                dZt = _empty((c_nc, 9, cout_s), T, dev) if want_w else None
                K.cond_head_bwd(c_counts, c_feat0, c_cond0, dzt, dF, dZt, c_s, c_nc)
        # passes whose GEMMs still need their own 1/sigma (spectral norm without BatchNorm: the story critic's first conv,
        # the heads' last conv): one set of launches per pass, on that pass's rows
        per_pass = sig is not None and not folded and ng > 1
        alpha1 = sig[0][1:] if (sig is not None and not folded) else None       # single-pass scale
        spans = _cum(counts, 1)
        out_w = {}

        def weight_side(side=None):
            """bias and weight gradients: everything that feeds only the optimizer (`side`: the stream it runs on when
            that is not the backward's own)"""
            dbias = dw = None
            # small fp32 dense layers over <= 64 rows: cpcsv_dense_rows_wgrad below also sums the columns of dz (the bias gradient)
            rows_wg = (_DENSE_ROWS and want_w and not ctx.conv and mod.slices == 1 and mod.tapmap is None and sig is None and not mod.fused
                       and not ctx.thin and dt == L.F32 and m <= 64 and ng == 1 and direct(weight) and weight.grad.is_contiguous()
                       and not getattr(weight, "_cpcsv_retired", False) and dzt.dtype == torch.float32 and x.dtype == torch.float32)
            db_fused = rows_wg and bias is not None and ctx.needs_input_grad[2] and direct(bias) and dzt is dz
            if bias is not None and ctx.needs_input_grad[2] and not db_fused:
                if direct(bias):
                    K.colsum(dz, bias.grad, m, cout, cout_s)            # accumulates straight into the flat grad buffer
                else:
                    dbias = _empty((cout,), torch.float32, dev, zero=True)
                    K.colsum(dz, dbias, m, cout, cout_s)
            # ---- weight gradient ----
            if want_w:
                g = mod.wgrad_buffer(dev)        # persistent fp32 accumulator: zero on entry, re-zeroed by unpack
                fused = mod.fused and direct(weight) and dt == mod.fused_dt
                if mod.fused and not fused:
                    raise RuntimeError("%s: its weight is on the deferred-update path (no master-layout .grad exists) but this "
                                       "call cannot use it (compute dtype changed after GANTrainer.setup?)" % mod.name)
                if fused and mod.fused_updated:
                    raise RuntimeError("%s: weight-gradient call after this step's in-backward update already ran (%d calls were "
                                       "expected per step); set CPCSV_INLINE_UPDATE=0 for irregular call patterns" % (mod.name, mod.fused_expected))

                def wdesc(xshape):
                    key = ("wgrad", xshape, dt)
                    wd = mod.descs.get(key)
                    if wd is None:
                        if ctx.sub:
                            n, ih, iw, cs = xshape
                            tiles = ((cout + 127) // 128) * ((cs + 127) // 128) * 16
                            wd = K.wgrad_desc(dtype=dt, M=n * ih * iw, N=cout, Cs=cs, ldy=cout_s, lddw=g.shape[1],
                                              taps=SUB_WGRAD_TAPS, MH=ih, MW=iw, IH=ih, IW=iw, splits=_splits_for(tiles, n * ih * iw),
                                              dy_gather=(2 * ih, 2 * iw, 2, 2), algo_scale=2.25)
                        elif ctx.conv:
                            n, ih, iw, cs = xshape
                            oh, ow = mod.geom.out_hw(ih, iw)
                            tiles = ((cout + 127) // 128) * ((cs + 127) // 128) * mod.slices
                            wd = K.wgrad_desc(dtype=dt, M=n * oh * ow, N=cout, Cs=cs, ldy=cout_s, lddw=g.shape[1],
                                              taps=mod.geom.fwd_taps(), MH=oh, MW=ow, IH=ih, IW=iw, sy=mod.geom.sh, sx=mod.geom.sw,
                                              up=mod.geom.up, splits=_splits_for(tiles, n * oh * ow))
                            if cond_t is not None:      # feature channels only: the accumulator keeps the layer's full K slices
                                wd.wstride = mod.cin_s
                                wd._algo = float(m * mod.cin_s) / float(n * oh * ow * cs)
                        else:
                            rows, cs = xshape
                            tiles = ((cout + 127) // 128) * ((cs + 127) // 128)
                            wd = K.wgrad_desc(dtype=dt, M=rows, N=cout, Cs=cs, ldy=cout_s, lddw=g.shape[1], taps=[(0, 0, 0)],
                                              splits=_splits_for(tiles, rows))
                        mod.descs[key] = wd
                    return wd

                def cond_columns(accumulate):
                    """the condition channels' columns of the accumulator: dW[o][tap][Cf + e] = sum_c dZt[c][tap][o] cond[c][e]"""
                    key = ("wgrad_cond", c_nc, dt)
                    wc = mod.descs.get(key)
                    if wc is None:
                        wc = mod.descs[key] = K.wgrad_desc(dtype=dt, M=c_nc, N=cout, Cs=c_es, ldy=9 * cout_s, lddw=g.shape[1],
                                                           taps=[(0, 0, t) for t in range(9)], splits=1, algo_scale=0.0)
                        wc.wstride, wc.dy_tapstride = mod.cin_s, cout_s
                    K.wgrad_run(wc, dZt, cond_t, g[:, c_cf:], accumulate=accumulate)

                # the launches: (dz rows, x rows, scale) - ONE over everything unless the passes carry their own 1/sigma
                if cond_t is not None:
                    parts = [(dF, x, None, None)]
                elif per_pass:
                    parts = [(dzt[spans[k]:spans[k + 1]], x[spans[k]:spans[k + 1]], sig[k][1:], k) for k in range(ng)]
                else:
                    parts = [(dzt, x, alpha1, None)]
                dots = []
                direct_done = False
                for dzp, xp, alpha, k in parts:
                    xs = tuple(xp.shape)
                    wd = wdesc(xs)
                    later = mod.fused_seen > 0 or (k is not None and k > 0)
                    if ctx.thin == 1:
                        n, ih, iw, cs = xs
                        slabs = mod.descs.get(("thin_slabs", xs))
                        if slabs is None:
                            slabs = mod.descs[("thin_slabs", xs)] = torch.empty(
                                K.thin3x3_wgrad_slabs(n, ih, iw, cs) * cout * 9 * cs, dtype=torch.float32, device=dev)
                        K.thin3x3_wgrad(dzp, xp, g, slabs, n, ih, iw, cs, cout)
                    elif ctx.thin == 2 and not fused and not per_pass and _thin4_slabs(mod, xs, cout, dev) is not None:
                        n, ih, iw, cs = xs
                        K.thin4x4s2_wgrad(dzp, xp, g, _thin4_slabs(mod, xs, cout, dev), n, ih, iw, cout)
                    elif fused:
                        # deferred update: this call only ADDS (already divided by its sigma) to the accumulator; unpack, Adam and
                        # the operand re-pack happen once per step in cpcsv_layer_update (cpcsv.optim.FusedAdam)
                        # (the step's FIRST call stores instead of adding: no read of the accumulator at all)
                        rows = wd.M
                        if not later:
                            # first write of the step into this layer's accumulator: a one-slice launch STORES, a pixel-split one
                            # adds with float atomics and needs zeros there. The gradient bucket skips the fill of store-first
                            # layers (cpcsv.dist.GradBucket.zero); should this launch need zeros it did not get, fill here.
                            need = wd.splits > 1 and not _runtime.deterministic()
                            if need and not getattr(mod, "_g_zeroed", True):
                                K.fill_zero(g)
                                mod._g_zeroed = True
                            mod.store_first = not need
                        pair = (_PAIR and mod.fused_expected == 2 and sig is None and dt == L.BF16 and cout > 64 and xp.shape[-1] > 64
                                and rows % 64 == 0 and rows % max(1, wd.MH * wd.MW) == 0 and ng == 1)
                        if pair and mod.fused_seen == 0:
                            # the story half and the image half of a generator pass share ONE weight-gradient launch: the first
                            # pass only parks its operands (kept alive here until that launch has been enqueued)
                            mod.fused_stash = (dzp, xp, wd, g)
                        elif pair and mod.fused_stash is not None and mod.fused_stash[1].shape == xp.shape:
                            dz1, x1 = mod.fused_stash[0], mod.fused_stash[1]
                            key2 = ("wgrad2", xs, dt)
                            wd2 = mod.descs.get(key2)
                            if wd2 is None:
                                wd2 = mod.descs[key2] = type(wd).from_buffer_copy(wd)   # same geometry, twice the rows
                                wd2.M = 2 * rows
                                wd2._algo = getattr(wd, "_algo", 1.0)
                            K.wgrad_run(wd2, dz1, x1, g, accumulate=0, second=(dzp, xp))
                            if side is not None:
                                keep_alive(dz1, x1)
                            mod.fused_stash = None
                        else:
                            flush_stash(mod)
                            K.wgrad_run(wd, dzp, xp, g, alpha=alpha, accumulate=1 if later else 0)
                            if cond_t is not None:
                                cond_columns(1 if later else 0)
                    elif (not ctx.conv and mod.slices == 1 and mod.tapmap is None and sig is None and direct(weight)
                          and weight.grad.is_contiguous() and not getattr(weight, "_cpcsv_retired", False)):
                        # small dense layers (text / motion encoders, GRU): the master [Cout][Cin] IS the accumulator layout minus
                        # the channel pads - add straight into the flat gradient buffer, no unpack launch
                        if rows_wg:
                            if _runtime.small_wgrads_deferred():      # parked: one launch for all of them at the end of the backward pass
                                _runtime.park_small_wgrad(weight.grad, bias.grad if db_fused else None, dzp, xp, xs[0], cout, mod.cin)
                            else:
                                K.dense_rows_wgrad(dzp, xp, weight.grad, xs[0], cout, mod.cin, bias.grad if db_fused else None)   # no atomics
                            direct_done = True
                            continue
                        key = ("wgrad_direct", xs, dt)
                        wdd = mod.descs.get(key)
                        if wdd is None:
                            wdd = mod.descs[key] = type(wd).from_buffer_copy(wd)
                            wdd.lddw, wdd.creal, wdd.splits = mod.cin, mod.cin, wd.splits
                            wdd._algo = getattr(wd, "_algo", 1.0)
                        K.wgrad_run(wdd, dzp, xp, weight.grad, accumulate=1)
                        direct_done = True
                    elif per_pass:
                        # one accumulator, several sigmas: each pass adds its share already divided by its sigma; its
                        # <G_k, W> (rank-1 term) needs the pass's OWN product, so it goes through a scratch accumulator
                        scratch = mod.descs.get(("wgrad_scratch", dt))
                        if scratch is None:
                            scratch = mod.descs[("wgrad_scratch", dt)] = torch.zeros_like(g)
                        if ctx.thin == 2 and _thin4_slabs(mod, xs, cout, dev) is not None:
                            K.thin4x4s2_wgrad(dzp, xp, scratch, _thin4_slabs(mod, xs, cout, dev), xs[0], xs[1], xs[2], cout)
                        else:
                            K.wgrad_run(wd, dzp, xp, scratch)
                        gwk = _empty((1,), torch.float32, dev)
                        K.wgrad_dot(scratch, weight, gwk, cout, mod.cin, mod.taps, mod.slices, mod.tapmap, mod.cin_s)
                        dots.append(gwk)
                        if direct(weight):
                            K.unpack_wgrad(scratch, weight.grad, sig[k], us[k], vs[k], gwk, cout, mod.cin, mod.taps, mod.slices, mod.tapmap,
                                           mod.cin_s, True)
                        else:
                            if dw is None:
                                dw = _empty_like(weight).zero_()
                            K.unpack_wgrad(scratch, dw, sig[k], us[k], vs[k], gwk, cout, mod.cin, mod.taps, mod.slices, mod.tapmap,
                                           mod.cin_s, True)
                    else:
                        K.wgrad_run(wd, dzp, xp, g)
                        if cond_t is not None:
                            cond_columns(0)
                if fused:
                    mod.fused_seen += 1
                    if sig is not None:                 # -(<G, W>/sigma^2) u v^T of every pass of THIS call, applied by the fused update
                        if gw_bn is None:
                            raise RuntimeError("%s: deferred update needs the closed-form <G,W> of a train-mode BatchNorm" % mod.name)
                        from . import modules as M_
                        for k in range(ng):
                            term = (gw_bn[k:k + 1], sig[k], us[k], vs[k])
                            mod.fused_terms.append(term)
                            if M_.TERM_LOG is not None:
                                M_.TERM_LOG.append((mod, term))
                elif not per_pass and not direct_done:
                    gws = None
                    if sig is not None:                     # d sigma / dW enters as -(<G, W>/sigma^2) u v^T per pass
                        if ctx.has_bn and mod.bn.training:
                            gws = [gw_bn[k:k + 1] for k in range(ng)]
                        else:
                            gw = _empty((1,), torch.float32, dev)
                            K.wgrad_dot(g, weight, gw, cout, mod.cin, mod.taps, mod.slices, mod.tapmap, mod.cin_s)
                            gws = [gw]
                    if ctx.sub:
                        if direct(weight):
                            K.unpack_wgrad_sum(g, weight.grad, cout, mod.cin, 9, 16, SUB_MASKS, mod.cin_s, True)
                        else:
                            dw = _empty_like(weight)
                            K.unpack_wgrad_sum(g, dw, cout, mod.cin, 9, 16, SUB_MASKS, mod.cin_s, False)
                    else:
                        tgt = weight.grad if direct(weight) else _empty_like(weight)
                        if folded:
                            # the accumulator already holds sum_k G_k / sigma_k: plain unpack, then the passes' rank-1 terms
                            K.unpack_wgrad(g, tgt, None, None, None, None, cout, mod.cin, mod.taps, mod.slices, mod.tapmap, mod.cin_s, direct(weight))
                            if us is not None and us[0] is not None:
                                for k in range(ng):
                                    K.rank1_sub(tgt, gws[k], sig[k], us[k], vs[k])
                        else:
                            sg0 = sig[0] if sig is not None else None
                            K.unpack_wgrad(g, tgt, sg0, us[0] if us is not None else None, vs[0] if vs is not None else None,
                                           gws[0] if gws is not None else None, cout, mod.cin, mod.taps, mod.slices, mod.tapmap, mod.cin_s,
                                           direct(weight))
                        if not direct(weight):
                            dw = tgt
            out_w["dw"], out_w["dbias"] = dw, dbias

        wside = wgrad_stream()
        inplace = (not want_w or direct(weight)) and (bias is None or not ctx.needs_input_grad[2] or direct(bias))
        # (the small fp32 layers' weight side is ONE ~5 us launch: a fork + join per layer costs the chain more than running it inline)
        tiny = _ROWS_INLINE and _DENSE_ROWS and not ctx.conv and dt == L.F32 and m <= 64 and not mod.fused and sig is None
        if wside is not None and inplace and not tiny and (want_w or (bias is not None and ctx.needs_input_grad[2])):
            fork_to(wside)                               # dz (and everything before it) is ordered before the side work
            with forced_stream(wside):
                weight_side(wside)
            keep_alive(dz, dzt, x, gw_bn, *(sig or ()), *(us or ()), *(vs or ()))   # main-pool tensors read over there: alive until the join
            if cond_t is not None:
                keep_alive(dF, dZt, cond_t)
            if not mod.fused:
                # master-layout gradients written over there (read-modify-write): a LATER inline pass of the same layer - the
                # <= 64-row half when the other half has more rows, ST*T <= 64 < IM or ST <= 64 < ST*T - must not overtake it
                note_side_write(wside, weight if want_w else None, bias)
        else:
            if wside is not None:
                wait_side_writes(weight, bias)
            weight_side()
        dw, dbias = out_w["dw"], out_w["dbias"]
        # ---- data gradient ----
        if ctx.needs_input_grad[0]:
            dx = _empty(ctx.xshape, T, dev, zero=bool(ctx.conv and not ctx.sub and not ctx.thin and not mod.geom.dgrad_covers_all()))
            if cond_t is not None:
                pieces = [(dF, dx, None)]
            elif per_pass:
                pieces = [(dzt[spans[k]:spans[k + 1]], dx[spans[k]:spans[k + 1]], sig[k][1:]) for k in range(ng)]
            else:
                pieces = [(dzt, dx, alpha1)]
            for dzp, dxp, alpha in pieces:
                xs = tuple(dxp.shape)
                if ctx.sub:
                    n, ih, iw, cs = xs
                    key = ("dgrad", 0, xs, dt, ctx.branch)
                    d = mod.descs.get(key)
                    if d is None:   # one stride-2 4x4 gather over dY with the summed weights
                        d = mod.descs[key] = K.gemm_desc(None, None, None, dtype=dt, M=n * ih * iw, N=mod.cin, Cs=cout_s,
                                                         ldb=bwd.shape[1], ldc=cs, taps=SUB_DGRAD_TAPS, MH=ih, MW=iw,
                                                         IH=2 * ih, IW=2 * iw, sy=2, sx=2)
                        d._algo = 2.25
                    K.bind(d, dzp, bwd, dxp, alpha)
                    ws = K.gemm_nt_auto(d, n * ih * iw, dev)
                    K.gemm_nt(d)
                    del ws
                elif ctx.thin == 1:
                    n, ih, iw, cs = xs
                    K.thin3x3_dgrad(dzp, bwd, dxp, n, ih, iw, cs, cout)
                elif ctx.thin == 2 and xs[2] == 64 and _THIN4_DGRAD:
                    n, ih, iw, cs = xs
                    K.thin4x4s2_dgrad(dzp, bwd, dxp, alpha, n, ih, iw)
                elif ctx.conv:
                    n, ih, iw, cs = xs
                    oh, ow = mod.geom.out_hw(ih, iw)
                    launches = mod.geom.dgrad_launches(ih, iw)
                    if len(launches) > 1 and len(launches) <= 4 and len({(l[1], l[2]) for l in launches}) == 1 \
                            and sum(len(l[0]) for l in launches) <= L.MAX_TAPS:
                        # transposed-conv parity phases of equal size: ONE launch, phase = blockIdx.z
                        taps, phases = [], []
                        for tp, mh, mw, _, sc in launches:
                            phases.append((len(taps), len(tp), sc[4], sc[5]))
                            taps += tp
                        sc0 = launches[0][4]
                        launches = [(taps, launches[0][1], launches[0][2], 0, (sc0[0], sc0[1], sc0[2], sc0[3], 0, 0), phases)]
                    for li, item in enumerate(launches):
                        taps, mh, mw, pool, scatter = item[:5]
                        phases = item[5] if len(item) > 5 else None
                        key = ("dgrad", li, xs, dt, ctx.branch)
                        d = mod.descs.get(key)
                        if d is None:
                            # dgrad_cols: only the first columns of dX are ever read (D_GET_LOGITS: the condition channels of its
                            # concatenated input are detached, reference model.py:89-92 / miscc/utils.py:69) - the rest of the
                            # row is not computed (pad columns of the last tile are written as zeros, the others stay unwritten)
                            d = mod.descs[key] = K.gemm_desc(None, None, None, dtype=dt, M=n * mh * mw, N=getattr(mod, "dgrad_cols", None) or mod.cin,
                                                             Cs=cout_s, ldb=bwd.shape[1], ldc=cs, taps=taps, MH=mh, MW=mw, IH=oh, IW=ow,
                                                             pool=pool, scatter=scatter, phases=phases)
                            if cond_t is not None:
                                d._algo = float(m) / float(n * mh * mw)      # (metering: the literal form runs this GEMM over every call's rows)
                        K.bind(d, dzp, bwd, dxp, alpha)
                        rows_out = n * ih * iw if scatter is not None else (n * mh * mw // 4 if pool else n * mh * mw)
                        ws = K.gemm_nt_auto(d, rows_out, dev)
                        K.gemm_nt(d)
                        del ws
                else:
                    rows, ks = xs
                    if (_DENSE_ROWS and dt == L.F32 and rows <= 64 and dzp.dtype == torch.float32 and dxp.dtype == torch.float32
                            and lin.shape[0] == ks):
                        # dX[m][i] = alpha * sum_o dz[m][o] * W[o][i]: the same one-launch product over the transposed operand copy
                        K.dense_rows(dzp, lin, dxp, rows, ks, cout_s, alpha, None, 0)
                        continue
                    key = ("dgrad", 0, xs, dt, ctx.branch)
                    d = mod.descs.get(key)
                    if d is None:
                        d = mod.descs[key] = K.gemm_desc(None, None, None, dtype=dt, M=rows, N=ks, Cs=cout_s, ldb=cout_s, ldc=ks,
                                                         taps=[(0, 0, 0)])
                    K.bind(d, dzp, lin, dxp, alpha)
                    ws = K.gemm_nt_auto(d, rows, dev)
                    K.gemm_nt(d)
                    del ws
        # ---- deferred update, in-backward form: this was the layer's last weight-gradient call of the step, so its fused
        # optimiser launch can go out NOW - after the data-gradient GEMM above has been enqueued (it reads the operand copy
        # the update rewrites) - on the weight-gradient branch, where it overlaps the rest of the backward chain instead of
        # queueing up behind it at optimizer.step()
        if mod.fused and mod.fused_expected and mod.fused_seen == mod.fused_expected and mod.fused_opt is not None \
                and mod.fused_opt.inline_ok():
            if late_stream() is not None and getattr(mod, "late_update", False):
                defer_late(lambda mod=mod: mod.fused_opt.update_layer_now(mod))
            elif wside is not None:
                fork_to(wside)
                with forced_stream(wside):
                    mod.fused_opt.update_layer_now(mod)
            else:
                mod.fused_opt.update_layer_now(mod)
        if getattr(mod, "late_flush", False):
            flush_late(wside)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None, None


class LogitHeadFn(Function):
    """D_GET_LOGITS' last layer - Conv2d(8*ndf, 1, 4, 4) + Sigmoid over the 4x4 map (reference model.py:79-80), spectral-normed,
    biased - as one launch forward and three backward (csrc/head.hip) instead of ~25 through LayerFn. x: [R, 16*Cin_s] (the
    flattened NHWC map); `groups`: row counts of the reference calls sharing the launch (real / wrong / fake), each with its own
    (sigma, u, v). Returns the probabilities [R, 1] fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, sigma, u, v, mod, groups):
        require_gpu(x)
        x = x.contiguous()
        r, kdim = x.shape
        dt = dcode(x)
        fwd, _, _ = mod.packs(weight, dt, "fwd")
        counts = tuple(int(c) for c in groups) if groups is not None and len(groups) > 1 else (r,)
        ng = len(counts)
        ctx.rows = _cum(counts, 1)
        ctx.sig, ctx.us, ctx.vs = _as_list(sigma, ng), _as_list(u, ng), _as_list(v, ng)
        p = _empty((r, 1), torch.float32, x.device)
        K.logit_head_fwd(x, fwd, bias, p, r, kdim, K.logit_groups(ctx.rows, ctx.sig))
        ctx.mod, ctx.dt = mod, dt
        if any(ctx.needs_input_grad[1:3]):
            # the weight-gradient scratch of the backward pass is created HERE, i.e. by the eager warm-up calls, never first inside
            # a graph capture (it would then live in that graph's private pool while eager fallbacks reuse it)
            key = ("logit_scratch", kdim, ng)
            if key not in mod.descs:
                mod.descs[key] = torch.empty(K.logit_head_scratch(kdim, ng), dtype=torch.float32, device=x.device)
        ctx.save_for_backward(x, weight, bias, p)
        return p

    @staticmethod
    @once_differentiable
    def backward(ctx, dp):
        x, weight, bias, p = ctx.saved_tensors
        mod, dt = ctx.mod, ctx.dt
        r, kdim = x.shape
        dev = x.device
        fwd, _, _ = mod.packs(weight, dt, "fwd")
        dy = dp.contiguous().float()
        dz = _empty((r,), torch.float32, dev)
        dx = _empty_like(x) if ctx.needs_input_grad[0] else None
        want_w, want_b = ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
        groups = K.logit_groups(ctx.rows, ctx.sig, ctx.us if want_w else None, ctx.vs if want_w else None)
        K.logit_head_bwd(dy, p, fwd, dx, dz, dt, r, kdim, groups)
        dw = db = None
        if want_w or want_b:
            direct = lambda q: q is not None and getattr(q, "_cpcsv_direct", False) and q.grad is not None and q.grad.is_contiguous()
            gw = weight.grad if direct(weight) else torch.zeros_like(weight)
            gb = (bias.grad if direct(bias) else torch.zeros_like(bias)) if want_b else None
            key = ("logit_scratch", kdim, len(ctx.rows) - 1)
            scratch = mod.descs.get(key)
            if scratch is None:
                scratch = mod.descs[key] = torch.empty(K.logit_head_scratch(kdim, len(ctx.rows) - 1), dtype=torch.float32, device=dev)
            K.logit_head_wgrad(dz, x, fwd, scratch, gw, gb, r, kdim, mod.cin, mod.cin_s, mod.slices, groups)
            dw = None if direct(weight) else gw
            db = None if (gb is None or direct(bias)) else gb
        return dx, dw, db, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# layout / glue
# ------------------------------------------------------------------------------------------------
class PadCastFn(Function):
    """fp32 pieces [B,k_i] -> one [B,Ks] matrix in the compute dtype (torch.cat + pad + cast)."""

    @staticmethod
    def forward(ctx, dtype, *xs):
        xs = [x.contiguous() for x in xs]
        require_gpu(xs[0])
        b = xs[0].shape[0]
        widths = [x.shape[1] for x in xs]
        ks = pad8(sum(widths))
        if len(xs) <= 4 and all(x.dtype == torch.float32 for x in xs):
            out = _empty((b, ks), dtype, xs[0].device)
            K.concat_pad(xs, out, b, ks)                    # cat + zero pad + cast in ONE launch
        else:
            out = _empty((b, ks), dtype, xs[0].device, zero=(ks != sum(widths)))
            col = 0
            for x, w in zip(xs, widths):
                K.copy2d(x, w, 0, out, ks, col, b, w)
                col += w
        ctx.widths, ctx.ks = widths, ks
        ctx.in_dtypes = [x.dtype for x in xs]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        b = dy.shape[0]
        grads, col = [], 0
        for i, w in enumerate(ctx.widths):
            if ctx.needs_input_grad[1 + i]:
                g = _empty((b, w), ctx.in_dtypes[i], dy.device)
                K.copy2d(dy, ctx.ks, col, g, w, 0, b, w)
                grads.append(g)
            else:
                grads.append(None)
            col += w
        return (None,) + tuple(grads)


class UnpadFn(Function):
    """columns [col0, col0+n) of a [B,Ns] matrix (any compute dtype) -> contiguous fp32 [B,n]."""

    @staticmethod
    def forward(ctx, y, col0, n):
        y = y.contiguous()
        b, ns = y.shape
        ctx.ns, ctx.n, ctx.col0, ctx.dtype = ns, n, col0, y.dtype
        out = _empty((b, n), torch.float32, y.device)
        K.copy2d(y, ns, col0, out, n, 0, b, n)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        dout = dout.contiguous()
        b = dout.shape[0]
        dy = _empty((b, ctx.ns), ctx.dtype, dout.device)
        K.copy2d(dout, ctx.n, 0, dy, ctx.ns, ctx.col0, b, ctx.n, fill=True)        # zeros outside the window, same launch
        return dy, None, None


def _planar_strides(x):
    """x is (F,C,H,W) or (B,C,T,H,W) with contiguous HxW planes -> (frames, T, sB, sT, sC, C, HW)."""
    if x.dim() == 4:
        f, c, h, w = x.shape
        sb, sc, sh, sw = x.stride()
        if sw != 1 or sh != w:
            return None
        return f, 1, sb, 0, sc, c, h * w
    b, c, t, h, w = x.shape
    sb, sc, st, sh, sw = x.stride()
    if sw != 1 or sh != w:
        return None
    return b * t, t, sb, st, sc, c, h * w


class ToNhwcFn(Function):
    """channel-planar fp32 images or stories -> NHWC frames in the compute dtype.
    (B,C,T,H,W) stories are unfolded to B*T frames like model.py:612-613 does."""

    @staticmethod
    def forward(ctx, x, dtype):
        require_gpu(x)
        geo = _planar_strides(x)
        if geo is None:
            x = x.contiguous()
            geo = _planar_strides(x)
        frames, t, sb, st, sc, c, hw = geo
        h, w = x.shape[-2], x.shape[-1]
        cs = pad8(c)
        out = _empty((frames, h, w, cs), dtype, x.device)
        K.planar_to_nhwc(x, out, frames, t, sb, st, sc, c, hw, cs)
        ctx.xshape, ctx.xdtype, ctx.c, ctx.cs = tuple(x.shape), x.dtype, c, cs
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = _empty(ctx.xshape, ctx.xdtype, dy.device)
        frames, t, sb, st, sc, c, hw = _planar_strides(dx)
        K.nhwc_to_planar(dy, dx, frames, t, sb, st, sc, c, hw, ctx.cs)
        return dx, None


class ToNhwcCatFn(Function):
    """Several channel-planar fp32 batches of the same frame shape -> ONE NHWC tensor, their frames back to back (the
    real and the fake batch of a critic update: the tower then runs once over both, cpcsv.runtime.row_groups)."""

    @staticmethod
    def forward(ctx, dtype, *xs):
        require_gpu(xs[0])
        geos, fixed = [], []
        for x in xs:
            geo = _planar_strides(x)
            if geo is None:
                x = x.contiguous()
                geo = _planar_strides(x)
            geos.append(geo)
            fixed.append(x)
        h, w = fixed[0].shape[-2], fixed[0].shape[-1]
        c = geos[0][5]
        cs = pad8(c)
        total = sum(g[0] for g in geos)
        out = _empty((total, h, w, cs), dtype, fixed[0].device)
        r0 = 0
        for x, (frames, t, sb, st, sc, cc, hw) in zip(fixed, geos):
            if cc != c or hw != h * w:
                raise RuntimeError("ToNhwcCatFn: pieces differ in channels / frame size")
            K.planar_to_nhwc(x, out[r0:r0 + frames], frames, t, sb, st, sc, c, hw, cs)
            r0 += frames
        ctx.meta = [(tuple(x.shape), x.dtype, g[0]) for x, g in zip(fixed, geos)]
        ctx.cs = cs
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        grads, r0 = [], 0
        for i, (shape, dtype, frames) in enumerate(ctx.meta):
            if ctx.needs_input_grad[1 + i]:
                dx = _empty(shape, dtype, dy.device)
                f, t, sb, st, sc, c, hw = _planar_strides(dx)
                K.nhwc_to_planar(dy[r0:r0 + frames], dx, f, t, sb, st, sc, c, hw, ctx.cs)
                grads.append(dx)
            else:
                grads.append(None)
            r0 += frames
        return (None,) + tuple(grads)


class ToPlanarFn(Function):
    """NHWC frames [N,H,W,Cs] -> fp32 NCHW [N,C,H,W] (the tensors the reference API returns)."""

    @staticmethod
    def forward(ctx, x, c):
        x = x.contiguous()
        n, h, w, cs = x.shape
        out = _empty((n, c, h, w), torch.float32, x.device)
        K.nhwc_to_planar(x, out, n, 1, c * h * w, 0, h * w, c, h * w, cs)
        ctx.c, ctx.shape, ctx.dtype = c, tuple(x.shape), x.dtype
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        geo = _planar_strides(dout)
        if geo is None:
            dout = dout.contiguous()
            geo = _planar_strides(dout)
        frames, t, sb, st, sc, c, hw = geo
        n, h, w, cs = ctx.shape
        dx = _empty(ctx.shape, ctx.dtype, dout.device)
        K.planar_to_nhwc(dout, dx, frames, t, sb, st, sc, c, hw, cs)
        return dx, None


class ToPlanarSplitFn(Function):
    """NHWC frames [N,H,W,Cs] holding several passes back to back -> one fp32 NCHW tensor per pass (row counts `counts`):
    the outputs of a generator pass that decoded its story half and its image half together."""

    @staticmethod
    def forward(ctx, x, c, counts):
        x = x.contiguous()
        n, h, w, cs = x.shape
        outs, r0 = [], 0
        for k in counts:
            out = _empty((k, c, h, w), torch.float32, x.device)
            K.nhwc_to_planar(x[r0:r0 + k], out, k, 1, c * h * w, 0, h * w, c, h * w, cs)
            outs.append(out)
            r0 += k
        ctx.c, ctx.shape, ctx.dtype, ctx.counts = c, tuple(x.shape), x.dtype, tuple(counts)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *douts):
        n, h, w, cs = ctx.shape
        dx = _empty(ctx.shape, ctx.dtype, douts[[d is not None for d in douts].index(True)].device)
        r0 = 0
        for k, dout in zip(ctx.counts, douts):
            part = dx[r0:r0 + k]
            if dout is None:
                K.fill_zero(part)
            else:
                geo = _planar_strides(dout)
                if geo is None:
                    dout = dout.contiguous()
                    geo = _planar_strides(dout)
                frames, t, sb, st, sc, c, hw = geo
                K.planar_to_nhwc(dout, part, frames, t, sb, st, sc, c, hw, cs)
            r0 += k
        return dx, None, None


class FeatToNhwcFn(Function):
    """[N, C*HW] features in (c,h,w) order (`.view(-1, C, 4, 4)`, model.py:379) -> NHWC [N,H,W,C]."""

    @staticmethod
    def forward(ctx, x, c, h, w):
        x = x.contiguous()
        n, ld = x.shape
        cs = pad8(c)
        out = _empty((n, h, w, cs), x.dtype, x.device)
        K.planar_to_nhwc(x, out, n, 1, ld, 0, h * w, c, h * w, cs)
        ctx.geo = (n, ld, c, h, w, cs)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, ld, c, h, w, cs = ctx.geo
        dy = dy.contiguous()
        dx = _empty((n, ld), dy.dtype, dy.device, zero=(ld != c * h * w))
        K.nhwc_to_planar(dy, dx, n, 1, ld, 0, h * w, c, h * w, cs)
        return dx, None, None, None


class Im2colFn(Function):
    """NHWC frames [F,H,W,Cs] -> patch matrix [F*OH*OW, ld] of a k x k / stride s / pad p conv over the first `c`
    channels, columns in the master weight's (c, ky, kx) order (cpcsv_im2col)."""

    @staticmethod
    def forward(ctx, x, c, k, s, p):
        x = x.contiguous()
        require_gpu(x)
        f, h, w, cs = x.shape
        oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        ld = pad8(c * k * k)
        out = _empty((f * oh * ow, ld), x.dtype, x.device)
        K.im2col(x, out, f, h, w, cs, c, k, s, p, ld)
        ctx.geo = (f, h, w, cs, c, k, s, p, ld)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dcol):
        f, h, w, cs, c, k, s, p, ld = ctx.geo
        dcol = dcol.contiguous()
        dx = _empty((f, h, w, cs), dcol.dtype, dcol.device)
        K.im2col(dcol, dx, f, h, w, cs, c, k, s, p, ld, adjoint=True)
        return dx, None, None, None, None


class GateFn(Function):
    """a*b + b  (model.py:383,387)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = _empty_like(a)
        K.gate_fwd(a, b, out)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        dout = dout.contiguous()
        da, db = _empty_like(a), _empty_like(b)
        K.gate_bwd(dout, a, b, da, db)
        return da, db


class MeanTFn(Function):
    """[N*T, ...] -> [N, ...] mean over T (model.py:616-617)."""

    @staticmethod
    def forward(ctx, x, t):
        x = x.contiguous()
        nt = x.shape[0]
        n = nt // t
        inner = x[0].numel()
        out = _empty((n,) + tuple(x.shape[1:]), x.dtype, x.device)
        K.mean_t(x, out, n, t, inner)
        ctx.geo = (n, t, inner, tuple(x.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        n, t, inner, shape = ctx.geo
        dout = dout.contiguous()
        dx = _empty(shape, dout.dtype, dout.device)
        K.mean_t_bwd(dout, dx, n, t, inner)
        return dx, None


class CondConcatFn(Function):
    """D_GET_LOGITS input (model.py:89-92): tile cond over the 4x4 map and concatenate on channels.
    cond is detached by the callers (miscc/utils.py:69,136) so it gets no gradient."""

    @staticmethod
    def forward(ctx, feat, cond, c):
        feat, cond = feat.contiguous(), cond.contiguous().float()
        n, h, w, cs_f = feat.shape
        e = cond.shape[1]
        cs_out = pad8(cs_f + e)
        out = _empty((n, h, w, cs_out), feat.dtype, feat.device)
        K.cond_concat(feat, cond, out, n, h * w, c, cs_f, e, cs_out)
        ctx.geo = (n, h, w, cs_f, cs_out, c)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        n, h, w, cs_f, cs_out, c = ctx.geo
        dout = dout.contiguous()
        df = _empty((n, h, w, cs_f), dout.dtype, dout.device)
        K.copy2d(dout, cs_out, 0, df, cs_f, 0, n * h * w, c, fill=True)
        return df, None, None


class CondTripletFn(Function):
    """The three D_GET_LOGITS inputs of a critic update (miscc/utils.py:74-84) as one tensor: feat = [real | fake] features
    (2N rows) -> [(real_i, cond_i) | (real_i, cond_{i+1}), i < N-1 | (fake_i, cond_i)] (3N-1 rows), cond tiled over the map."""

    @staticmethod
    def forward(ctx, feat, cond, c):
        feat, cond = feat.contiguous(), cond.contiguous().float()
        n2, h, w, cs_f = feat.shape
        n = n2 // 2
        e = cond.shape[1]
        cs_out = pad8(cs_f + e)
        out = _empty((3 * n - 1, h, w, cs_out), feat.dtype, feat.device)
        K.cond_triplet(feat, cond, out, n, h * w, c, cs_f, e, cs_out)
        ctx.geo = (n, h, w, cs_f, cs_out, c)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        n, h, w, cs_f, cs_out, c = ctx.geo
        dout = dout.contiguous()
        df = _empty((2 * n, h, w, cs_f), dout.dtype, dout.device)
        K.cond_triplet_bwd(dout, df, n, h * w, c, cs_f, cs_out)
        return df, None, None


# ------------------------------------------------------------------------------------------------
# recurrent pieces
# ------------------------------------------------------------------------------------------------
class GruPointFn(Function):
    """GRUCell gate math on gi = W_ih x + b_ih, gh = W_hh h + b_hh (fp32, row stride ldg). h may be [B, H] or the padded
    [B, Hs] the W_hh layer reads (pads zero): the new state has the same layout, so a recurrence never re-pads."""

    @staticmethod
    def forward(ctx, gi, gh, h, hdim):
        gi, gh, h = gi.contiguous(), gh.contiguous(), h.contiguous()
        b, ldg = gi.shape
        ldh = h.shape[1]
        hnew = _empty_like(h)
        gates = _empty((b, 4 * hdim), torch.float32, h.device)
        K.gru_gates_fwd(gi, gh, h, hnew, gates, b, hdim, ldg, ldh)
        ctx.save_for_backward(gates, h)
        ctx.geo = (b, hdim, ldg, ldh)
        return hnew

    @staticmethod
    @once_differentiable
    def backward(ctx, dh):
        gates, h = ctx.saved_tensors
        b, hdim, ldg, ldh = ctx.geo
        dh = dh.contiguous()
        dgi = _empty((b, ldg), torch.float32, h.device)        # the kernel zeroes the row pads
        dgh = _empty((b, ldg), torch.float32, h.device)
        dhp = _empty_like(h)
        K.gru_gates_bwd(dh, gates, h, dgi, dgh, dhp, b, hdim, ldg, ldh)
        return dgi, dgh, dhp, None


class GruSeqFn(Function):
    """A whole GRUCell recurrence (reference model.py:320-346: `for t: h = cell(x_t, h)`) from the input gates of all steps:
    gi_all [T,B,ldg] (= W_ih x_t + b_ih, one product over the time-major stack), h0 [B,ldh] (padded state layout, pads zero)
    -> the states [T,B,ldh]. Forward: one cpcsv_gru_step_fwd launch per step. Backward: per step the gate gradients
    (cpcsv_gru_gates_bwd) and dh_{t-1} = dh_t * z + dgh_t W_hh as ONE cpcsv_dense_rows launch (the direct path rides in as `init`);
    the weight and bias gradients of W_hh once for the whole sequence (the rows of all steps stacked: T*B <= 64 per launch) - instead
    of five launches and two autograd additions per step. `lay` is the cell's W_hh KernelLayer (operand copies)."""

    @staticmethod
    def forward(ctx, gi_all, h0, w_hh, b_hh, lay, hdim):
        gi_all, h0 = gi_all.contiguous(), h0.contiguous()
        require_gpu(gi_all)
        t_, b, ldg = gi_all.shape
        ldh = h0.shape[1]
        fwd, _, _ = lay.packs(w_hh, L.F32, "both" if (_EARLY_BWD_PACK and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])) else "fwd")
        hall = _empty((t_ + 1, b, ldh), torch.float32, h0.device)
        hall[0].copy_(h0)
        gates = _empty((t_, b, 4 * hdim), torch.float32, h0.device)
        for t in range(t_):
            K.gru_step_fwd(gi_all[t], hall[t], fwd, b_hh, hall[t + 1], gates[t], b, hdim)
        ctx.save_for_backward(gates, hall, w_hh, b_hh)
        ctx.lay, ctx.geo = lay, (t_, b, hdim, ldg, ldh)
        return hall[1:]

    @staticmethod
    @once_differentiable
    def backward(ctx, dhs):
        gates, hall, w_hh, b_hh = ctx.saved_tensors
        t_, b, hdim, ldg, ldh = ctx.geo
        lay = ctx.lay
        dev = hall.device
        _, _, lin = lay.packs(w_hh, L.F32, "bwd")                    # W_hh^T: [ldh][pad8(3H)]
        dgi = _empty((t_, b, ldg), torch.float32, dev)
        dgh = _empty((t_, b, ldg), torch.float32, dev)
        dhp = _empty((b, ldh), torch.float32, dev)
        acc = dhs.contiguous()                                        # acc[t] = d loss / d h_t: its own gradient + what step t+1 sends back
        if acc.data_ptr() == dhs.data_ptr():
            acc = dhs.clone()                                         # (never accumulate into the caller's tensor)
        dh0 = _empty((b, ldh), torch.float32, dev)
        for t in range(t_ - 1, -1, -1):
            K.gru_gates_bwd(acc[t], gates[t], hall[t], dgi[t], dgh[t], dhp, b, hdim, ldg, ldh)
            # dh_{t-1} (+)= dh_t * z (dhp) + dgh_t W_hh: one launch, added to the gradient the state already has as an output
            K.dense_rows(dgh[t], lin, acc[t - 1] if t > 0 else dh0, b, ldh, lin.shape[1], None, None, 0, None, 0, init=dhp,
                         accumulate=1 if t > 0 else 0)
        carry = dh0
        direct = lambda p: getattr(p, "_cpcsv_direct", False) and p.grad is not None
        dw = db = None
        want_w, want_b = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        if want_w or want_b:
            if direct(w_hh) and w_hh.grad.is_contiguous():
                gw = w_hh.grad
            else:
                gw = dw = torch.zeros_like(w_hh)
            if want_b and direct(b_hh):
                gb = b_hh.grad
            elif want_b:
                gb = db = torch.zeros_like(b_hh)
            else:
                gb = None
            steps_per = max(1, 64 // b)
            wait_side_writes(w_hh, b_hh)          # (the > 64-row pass of the same cell went through LayerFn's weight-gradient branch)
            park = _runtime.small_wgrads_deferred() and gw is w_hh.grad and (gb is None or gb is b_hh.grad)
            for t0 in range(0, t_, steps_per):
                t1 = min(t_, t0 + steps_per)
                if park:
                    _runtime.park_small_wgrad(gw, gb, dgh[t0:t1].reshape(-1, ldg), hall[t0:t1].reshape(-1, ldh), (t1 - t0) * b, 3 * hdim, hdim)
                else:
                    K.dense_rows_wgrad(dgh[t0:t1].reshape(-1, ldg), hall[t0:t1].reshape(-1, ldh), gw, (t1 - t0) * b, 3 * hdim, hdim, gb)
        return dgi, carry, dw, db, None, None


class DynFilter1dFn(Function):
    """DynamicFilterLayer1D.forward (layers.py:69-80) as one launch: sig (N,C,L), taps (N,1,C,K) -> (N,1,L)."""

    @staticmethod
    def forward(ctx, sig, taps, pad):
        sig, taps = sig.contiguous(), taps.contiguous()
        require_gpu(sig)
        n, c, ln = sig.shape
        k = taps.shape[-1]
        out = _empty((n, 1, ln), torch.float32, sig.device)
        K.dfl1d_fwd(sig, taps, out, n, c, ln, k, pad)
        ctx.save_for_backward(sig, taps)
        ctx.geo = (n, c, ln, k, pad)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        sig, taps = ctx.saved_tensors
        n, c, ln, k, pad = ctx.geo
        dout = dout.contiguous()
        dsig, dtaps = _empty_like(sig), _empty_like(taps)
        K.dfl1d_bwd(dout, sig, taps, dsig, dtaps, n, c, ln, k, pad)
        return dsig, dtaps, None


class ReparamFn(Function):
    """eps * exp(0.5*logvar) + mu (model.py:53-60)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        out = _empty_like(mu)
        K.reparam_fwd(mu, logvar, eps, out)
        ctx.save_for_backward(logvar, eps)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        logvar, eps = ctx.saved_tensors
        dout = dout.contiguous()
        dmu, dlv = _empty_like(dout), _empty_like(dout)
        K.reparam_bwd(dout, logvar, eps, dmu, dlv)
        return dmu, dlv, None


# ------------------------------------------------------------------------------------------------
# scalar losses: forward computes the loss AND its local gradient; backward scales by the upstream
# ------------------------------------------------------------------------------------------------
def _chain(local, upstream):
    out = _empty_like(local)
    K.scale_by(local, out, upstream.contiguous().float())
    return out


class BceFn(Function):
    """nn.BCELoss on probabilities (miscc/utils.py:51)."""

    @staticmethod
    def forward(ctx, p, target):
        p, target = p.contiguous(), target.contiguous()
        loss = _empty((1,), torch.float32, p.device)
        grad = _empty_like(p)
        K.bce_fwd(p, target, loss, grad)
        ctx.save_for_backward(grad)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _chain(grad, g.reshape(1)), None


class BceGroupsFn(Function):
    """errD_real + 0.5 * (errD_wrong + errD_fake) of a critic update (miscc/utils.py:76-101) from the concatenated
    [real | wrong | fake] probabilities in one launch. Returns (weighted total (differentiable), the three means (detached))."""

    @staticmethod
    def forward(ctx, p, target, sizes, weights):
        p, target = p.contiguous(), target.contiguous()
        out = _empty((4,), torch.float32, p.device)
        grad = _empty_like(p)
        K.bce_groups(p, target, out, grad, sizes[0], sizes[1], sizes[2], weights[0], weights[1], weights[2])
        ctx.save_for_backward(grad)
        parts = out[:3]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)        # (else autograd zero-fills a gradient for `parts` on every backward: a launch)
        return out[3], parts

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gparts):
        if g is None:
            return None, None, None, None
        (grad,) = ctx.saved_tensors
        return _chain(grad, g.reshape(1)), None, None, None


class MlsmFn(Function):
    """nn.MultiLabelSoftMarginLoss (miscc/utils.py:52); logits may be a padded [N, ld] matrix. Returns (loss, accuracy): the
    accuracy is get_multi_acc of the same logits (reference miscc/utils.py:108,153,313-321), a non-differentiable device scalar
    out of the same launch."""

    @staticmethod
    def forward(ctx, logits, target, c):
        logits, target = logits.contiguous(), target.contiguous()
        n, ld = logits.shape
        out = _empty((2,), torch.float32, logits.device)
        grad = _empty((n, ld), torch.float32, logits.device)   # the kernel zeroes the column pads
        K.mlsm_fwd(logits, target, out[0:1], grad, n, c, ld, acc=out[1:2])
        ctx.save_for_backward(grad)
        acc = out[1]
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)        # (no zero-filled gradient for `acc`)
        return out[0], acc

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gacc):
        if g is None:
            return None, None, None
        (grad,) = ctx.saved_tensors
        return _chain(grad, g.reshape(1)), None, None


class LinCombFn(Function):
    """sum_i w_i * x_i of up to 8 scalar tensors in ONE launch forward and one backward (the generator's total loss,
    reference trainer.py:409-413: otherwise a dozen scalar mul / add launches each way)."""

    @staticmethod
    def forward(ctx, weights, *xs):
        xs = [x.reshape(1).contiguous().float() for x in xs]
        lst = K.scalar_list(xs, weights)
        out = _empty((1,), torch.float32, xs[0].device)
        K.lincomb_fwd(lst, out)
        ctx.lst, ctx.keep = lst, xs
        return out.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n = ctx.lst.n
        dx = _empty((n,), torch.float32, g.device)
        K.lincomb_bwd(g.reshape(1).contiguous().float(), ctx.lst, dx)
        return (None,) + tuple(dx[i] for i in range(n))


class KlFn(Function):
    """KL_loss (miscc/utils.py:184-188)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        loss = _empty((1,), torch.float32, mu.device)
        dmu, dlv = _empty_like(mu), _empty_like(mu)
        K.kl_fwd(mu, logvar, loss, dmu, dlv)
        ctx.save_for_backward(dmu, dlv)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dmu, dlv = ctx.saved_tensors
        g = g.reshape(1)
        return _chain(dmu, g), _chain(dlv, g)


class MseFn(Function):
    """nn.MSELoss (trainer.py:222) on two same-shape tensors of the compute dtype or fp32."""

    @staticmethod
    def forward(ctx, a, b, count=0):
        a, b = a.contiguous(), b.contiguous()
        loss = _empty((1,), torch.float32, a.device)
        da = _empty_like(a) if ctx.needs_input_grad[0] else None
        db = _empty_like(b) if ctx.needs_input_grad[1] else None
        K.mse_fwd(a, b, loss, da, db, count)
        ctx.save_for_backward(da, db)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        g = g.reshape(1)
        return (_chain(da, g) if da is not None else None), (_chain(db, g) if db is not None else None), None


def compute_dtype():
    return tdtype()
