"""Thin typed wrappers: torch tensors in, one C-ABI call out (include/cpcsv_hip.h).

No arithmetic happens here; every function launches exactly the HIP kernels the C entry
point names, on torch's current stream. Tensors must be contiguous CUDA(HIP) tensors.
"""
import ctypes as C
import os as _os

import torch

from . import _lib
from .runtime import dcode, ptr, stream

L = _lib


_FN = {}
# tools/ablate.sh only: entry points whose launches are SKIPPED (results are then wrong; the step time shows how much of
# the wall clock that kernel family really holds once stream overlap is accounted for)
_ABLATE = frozenset(x for x in _os.environ.get("CPCSV_ABLATE", "").split(",") if x)


def _call(name, *args):
    if _ABLATE and name in _ABLATE:
        return
    fn = _FN.get(name)
    if fn is None:                      # bound once: ~2000 launches per step go through here
        fn = _FN[name] = getattr(L.load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def make_taps(entries):
    arr = (L.Tap * L.MAX_TAPS)()
    for i, e in enumerate(entries):
        arr[i].oy, arr[i].ox, arr[i].wtap = e[0], e[1], e[2]
        if len(e) > 3:
            arr[i]._pad = e[3]
    return arr


def gemm_desc(A, B, Cout, *, dtype, M, N, Cs, ldb, ldc, taps, MH=1, MW=1, IH=1, IW=1, sy=1, sx=1, up=0, pool=0,
              scatter=None, alpha=None, bias=None, act=0, stats=None, ldstat=0, out_f32=0, phases=None):
    d = L.GemmDesc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(Cout)
    d.dtype, d.M, d.N, d.Cs, d.ldb, d.ldc = dtype, M, N, Cs, ldb, ldc
    d.ntaps = len(taps)
    d.taps = make_taps(taps)
    d.MH, d.MW, d.IH, d.IW, d.sy, d.sx = MH, MW, IH, IW, sy, sx
    d.up_shift, d.pool_rows = up, pool
    if scatter is not None:
        d.scatter = 1
        d.OH, d.OW, d.osy, d.osx, d.ooy, d.oox = scatter
    d.alpha, d.bias, d.act = ptr(alpha), ptr(bias), act
    d.stats, d.ldstat, d.out_f32 = ptr(stats), ldstat, out_f32
    # block order by which operand is bigger: input pixels (A) or packed weights (B)
    imgs = max(1, M // max(1, MH * MW))
    d.order_m_fast = int(N * ldb > imgs * IH * IW * Cs)
    if phases is not None:          # [(tap0, ntaps, ooy, oox)] parity phases folded into the block id
        d.nphases = len(phases)
        for i, (t0, nt, oy, ox) in enumerate(phases):
            d.ph_tap0[i], d.ph_ntaps[i], d.ph_ooy[i], d.ph_oox[i] = t0, nt, oy, ox
    return d


# split-K planner constants (swept in rounds 1-3 and again at round 6's HEAD, profiles/r06_knob_sweep.txt: nothing within +-0.1 ms moves;
# their environment overrides are retired - tests lower _SPLIT_MIN_NK / _SPLIT_MINK through these module attributes)
_SPLIT_TILES = 400
_SPLIT_BLOCKS = 480
_SPLIT_MINK = 16
_SPLIT_MIN_NK = 32
_SPLIT_LONGK = 96
_SKINNY_MIN_NK = 6
_SKINNY_SPLIT = 8


def plan_splitk(desc, k_tile):
    """Split-K factor for few-tile / long-K shapes (small maps with wide channels, dense layers with a tiny
    batch): enough blocks to fill 256 CUs a few times over, each still looping >= 4 K tiles."""
    m, n = desc.M, desc.N
    lib = L.load()
    keep, desc.splitk = desc.splitk, 1
    bm, bn = lib.cpcsv_gemm_mtile(C.byref(desc)), lib.cpcsv_gemm_ntile(C.byref(desc))   # the kernel's own tile choice
    desc.splitk = keep
    tiles = ((m + bm - 1) // bm) * ((n + bn - 1) // bn) * max(1, desc.nphases)
    ntaps = max(desc.ph_ntaps[:desc.nphases]) if desc.nphases > 1 else desc.ntaps
    nk = ntaps * ((desc.Cs + k_tile - 1) // k_tile)
    # A split costs a second launch (slab reduction) and 2 x splits x output bytes of fp32 traffic. Measured on the
    # critic shapes (tools/gemm_sweep.py, profiles/r01_gemm_by_shape.txt): aim for ~480 blocks, keep >= 16 K tiles per
    # slice, at most 8 slices unless the output is tiny (heads: a few KB), never below 32 K tiles in total.
    # tiny-batch dense layers (the text / motion encoders, GRU steps: M <= 64 rows): one block per 128 output columns
    # walks ALL K tiles with a global->LDS round trip each (~1.5 us): 24 us for a 30 MFLOP product, on the critical path of
    # every generator pass. Slices of 2-3 K tiles + the slab reduction bring it to the launch floor.
    if m <= 64 and nk >= _SKINNY_MIN_NK and tiles < 64:
        return int(max(1, min(nk // 2, _SKINNY_SPLIT)))
    if tiles >= _SPLIT_TILES or nk < _SPLIT_MIN_NK or (tiles >= 200 and nk < _SPLIT_LONGK):
        return 1
    out_bytes = 4 * m * n * max(1, desc.nphases)
    cap = 8 if out_bytes > (1 << 20) else 32
    return int(max(1, min((_SPLIT_BLOCKS + tiles - 1) // tiles, nk // _SPLIT_MINK, cap)))


def bind(desc, A, B, Cout, alpha=None, bias=None):
    """Point a cached descriptor at this call's buffers (geometry is static per layer and input shape)."""
    desc.A, desc.B, desc.C = ptr(A), ptr(B), ptr(Cout)
    desc.alpha, desc.bias = ptr(alpha), ptr(bias)
    desc.stats, desc.ldstat = None, 0
    return desc


def set_row_groups(desc, rows):
    """rows: cumulative GEMM-row offsets of the passes [0, ..., M] (cpcsv_gemm_desc.ngroups / grow)."""
    desc.ngroups = len(rows) - 1
    for i, r in enumerate(rows):
        desc.grow[i] = r


def bind_group_alpha(desc, alphas):
    for g in range(4):
        desc.galpha[g] = ptr(alphas[g]) if (alphas is not None and g < len(alphas)) else None


def gemm_nt_auto(desc, out_rows, dev):
    """gemm_nt with the split-K decision and workspace handled; returns the workspace (kept alive by the caller
    until the stream work is enqueued)."""
    sk = getattr(desc, "_sk", None)
    if sk is None:
        sk = plan_splitk(desc, 64 if desc.dtype == L.BF16 else 32)
        desc._sk = sk
    ws = None
    if sk > 1:
        ws = getattr(desc, "_ws", None)
        if ws is None or ws.device != dev:
            ldws = (desc.N + 7) // 8 * 8
            # persistent per descriptor; slabs are fully rewritten by every call, so no zeroing ever
            ws = desc._ws = torch.empty((sk, out_rows, ldws), dtype=torch.float32, device=dev)
            if _os.environ.get("CPCSV_POISON", "0") == "1":
                ws.fill_(float("nan"))
            desc.splitk, desc.ws, desc.ldws, desc.ws_rows = sk, ws.data_ptr(), ldws, out_rows
    return ws


def gemm_mtile(desc):
    return L.load().cpcsv_gemm_mtile(C.byref(desc))


def gemm_nt(desc):
    _call("cpcsv_gemm_nt", C.byref(desc), stream())


_WGRAD_LEGACY = int(_os.environ.get("CPCSV_WGRAD_LEGACY", "0"))


def wgrad_desc(*, dtype, M, N, Cs, ldy, lddw, taps, MH=1, MW=1, IH=1, IW=1, sy=1, sx=1, up=0, splits=1,
               dy_gather=None, algo_scale=1.0):
    """Static part of a weight-gradient launch (cached per layer and input shape by the caller).
    algo_scale: reference-algorithm FLOPs / executed FLOPs of this launch (metering only)."""
    d = L.WgradDesc()
    if dy_gather is not None:
        d.dy_gather = 1
        d.DYH, d.DYW, d.dy_sy, d.dy_sx = dy_gather
    d.dtype, d.M, d.N, d.Cs, d.ldy, d.lddw = dtype, M, N, Cs, ldy, lddw
    d.ntaps = len(taps)
    d.taps = make_taps(taps)
    d.MH, d.MW, d.IH, d.IW, d.sy, d.sx, d.up_shift, d.splits = MH, MW, IH, IW, sy, sx, up, splits
    d.legacy = _WGRAD_LEGACY
    d._algo = algo_scale
    return d


def wgrad_run(d, dY, X, dW, alpha=None, accumulate=0, second=None):
    """alpha / accumulate: deferred-update layers (cpcsv.optim.FusedAdam.attach_layer) fold 1/sigma into this call's
    contribution and add to whatever earlier calls of the step left in dW. second = (dY2, X2): the other pass of the
    same layer rides in the same launch (the descriptor's M covers both, M1 = rows of the first)."""
    d.dY, d.X, d.dW = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
    d.alpha, d.accumulate = ptr(alpha), int(accumulate)
    if second is not None:
        d.dY2, d.X2, d.M1 = second[0].data_ptr(), second[1].data_ptr(), d.M // 2
    else:
        d.dY2, d.X2, d.M1 = None, None, 0
    _call("cpcsv_wgrad_tn", C.byref(d), stream())


def layer_update(desc):
    _call("cpcsv_layer_update", C.byref(desc), stream())


def wgrad_tn(dY, X, dW, **kw):
    wgrad_run(wgrad_desc(**kw), dY, X, dW)


_ARRAYS = {}


def _tapmap(tapmap):
    """ctypes image of a tap map, built once per distinct map (a few per process)."""
    if tapmap is None:
        return None
    key = ("t",) + tuple(tapmap)
    arr = _ARRAYS.get(key)
    if arr is None:
        arr = _ARRAYS[key] = (C.c_int8 * len(tapmap))(*tapmap)
    return arr


def pack_weight(w, fwd, bwd, lin, dtype, Cout, Cin, taps, S, tapmap, Cin_s, Cout_s):
    tm = _tapmap(tapmap)
    _call("cpcsv_pack_weight", ptr(w), ptr(fwd), ptr(bwd), ptr(lin), dtype, Cout, Cin, taps, S,
          C.cast(tm, C.c_void_p) if tm is not None else None, Cin_s, Cout_s, stream())


def unpack_wgrad(G, dw, sigma, u, v, gw_dot, Cout, Cin, taps, S, tapmap, Cin_s, accumulate, rezero=1):
    tm = _tapmap(tapmap)
    _call("cpcsv_unpack_wgrad", ptr(G), ptr(dw), ptr(sigma), ptr(u), ptr(v), ptr(gw_dot), Cout, Cin, taps, S,
          C.cast(tm, C.c_void_p) if tm is not None else None, Cin_s, int(accumulate), int(rezero), stream())


def rank1_sub(dw, gw, sigma, u, v):
    _call("cpcsv_rank1_sub", ptr(dw), ptr(gw), ptr(sigma), ptr(u), ptr(v), u.numel(), v.numel(), stream())


def _masks(masks):
    key = ("m",) + tuple(masks)
    arr = _ARRAYS.get(key)
    if arr is None:
        arr = _ARRAYS[key] = (C.c_uint16 * len(masks))(*masks)
    return arr


def pack_weight_sum(w, fwd, bwd, dtype, Cout, Cin, taps, S, masks, Cin_s, Cout_s):
    mk = _masks(masks)
    _call("cpcsv_pack_weight_sum", ptr(w), ptr(fwd), ptr(bwd), dtype, Cout, Cin, taps, S, C.cast(mk, C.c_void_p), Cin_s,
          Cout_s, stream())


def unpack_wgrad_sum(G, dw, Cout, Cin, taps, S, masks, Cin_s, accumulate, rezero=1):
    mk = _masks(masks)
    _call("cpcsv_unpack_wgrad_sum", ptr(G), ptr(dw), Cout, Cin, taps, S, C.cast(mk, C.c_void_p), Cin_s, int(accumulate),
          int(rezero), stream())


def wgrad_dot(G, w, out, Cout, Cin, taps, S, tapmap, Cin_s):
    tm = _tapmap(tapmap)
    _call("cpcsv_wgrad_dot", ptr(G), ptr(w), ptr(out), Cout, Cin, taps, S,
          C.cast(tm, C.c_void_p) if tm is not None else None, Cin_s, stream())


def thin_supported(kind, Cs, Cout, H, W):
    return bool(L.load().cpcsv_thin_supported(kind, Cs, Cout, H, W))


def thin3x3_fwd(x, w_fwd, y, N, H, W, Cs, Cout, act):
    _call("cpcsv_thin3x3_fwd", ptr(x), ptr(w_fwd), ptr(y), N, H, W, Cs, Cout, act, stream())


def thin3x3_dgrad(dz, w_bwd, dx, N, H, W, Cs, Cout):
    _call("cpcsv_thin3x3_dgrad", ptr(dz), ptr(w_bwd), ptr(dx), N, H, W, Cs, Cout, stream())


def thin3x3_wgrad_slabs(N, H, W, Cs):
    return L.load().cpcsv_thin3x3_wgrad_slabs(N, H, W, Cs)


def thin3x3_wgrad(dz, x, G, slabs, N, H, W, Cs, Cout):
    _call("cpcsv_thin3x3_wgrad", ptr(dz), ptr(x), ptr(G), ptr(slabs), N, H, W, Cs, Cout, stream())


def thin4x4s2_dgrad(dz, w_bwd, dx, alpha, N, H, W):
    _call("cpcsv_thin4x4s2_dgrad", ptr(dz), ptr(w_bwd), ptr(dx), ptr(alpha), N, H, W, stream())


def thin4x4s2_wgrad_slabs(N, H, W):
    return L.load().cpcsv_thin4x4s2_wgrad_slabs(N, H, W)


def thin4x4s2_wgrad(dz, x, G, slabs, N, H, W, Cout):
    _call("cpcsv_thin4x4s2_wgrad", ptr(dz), ptr(x), ptr(G), ptr(slabs), N, H, W, Cout, stream())


def thin4x4s2_fwd(x, w_fwd, y, alpha, N, H, W, Cout, act):
    _call("cpcsv_thin4x4s2_fwd", ptr(x), ptr(w_fwd), ptr(y), ptr(alpha), N, H, W, Cout, act, stream())


def spectral_sigma(w, u, v, out, work, rows, cols, iterate, snapshot):
    _call("cpcsv_spectral_sigma", ptr(w), ptr(u), ptr(v), ptr(out), ptr(work), rows, cols, int(iterate), int(snapshot), stream())


def sn_multi_blocks(rows, cols, which):
    return L.load().cpcsv_sn_multi_blocks(rows, cols, which)


def spectral_sigma_multi1(jobs, njobs, start, nblk, part, part_off, max_rows):
    _call("cpcsv_spectral_sigma_multi1", ptr(jobs), njobs, ptr(start), nblk, ptr(part), ptr(part_off), int(max_rows), stream())


def spectral_sigma_multi(jobs, njobs, start1, nblk1, start2, nblk2, iterate):
    _call("cpcsv_spectral_sigma_multi", ptr(jobs), njobs, ptr(start1), nblk1, ptr(start2), nblk2, int(iterate), stream())


def bn_groups(rows, pstride, tiles=None, nph=1, sigmas=None):
    """cpcsv_bn_groups: `rows` = cumulative row offsets [0, ..., total]; `tiles` = cumulative statistics-partial counts
    (finalize); sigmas = per-group {sigma, 1/sigma} tensors (backward apply)."""
    g = L.BnGroups()
    g.n = len(rows) - 1
    for i, r in enumerate(rows):
        g.row[i] = r
    g.pstride = pstride
    if tiles is not None:
        for i, t in enumerate(tiles):
            g.tile[i] = t
        g.TM = tiles[-1]
    g.nph = nph
    if sigmas is not None:
        for i, sg in enumerate(sigmas):
            g.sigma[i] = ptr(sg)
    return g


def _gref(groups):
    return C.byref(groups) if groups is not None else None


def dense_rows(x, w, y, M, N, Kdim, alpha=None, bias=None, act=0, stats=None, ldstat=0, init=None, accumulate=0):
    _call("cpcsv_dense_rows", ptr(x), x.shape[1], ptr(w), w.shape[1], ptr(y), y.shape[1], M, N, Kdim, ptr(alpha), ptr(bias), act,
          ptr(stats), ldstat, ptr(init), init.shape[1] if init is not None else 0, int(accumulate), stream())


def gru_step_fwd(gi, h, w_hh, b_hh, hnew, gates, B, H):
    _call("cpcsv_gru_step_fwd", ptr(gi), gi.shape[1], ptr(h), h.shape[1], ptr(w_hh), w_hh.shape[1], ptr(b_hh), ptr(hnew), ptr(gates), B, H, stream())


def dense_rows_wgrad(dz, x, dW, M, N, Kr, db=None):
    _call("cpcsv_dense_rows_wgrad", ptr(dz), dz.shape[1], ptr(x), x.shape[1], ptr(dW), ptr(db), M, N, Kr, stream())


def bn_finalize(partials, mtiles, ldstat, count, gamma, beta, rmean, rvar, mean, invstd, scale, shift, Cn, Cs, eps,
                momentum, update, bwd_sums=None, groups=None):
    _call("cpcsv_bn_finalize", ptr(partials), mtiles, ldstat, count, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
          ptr(mean), ptr(invstd), ptr(scale), ptr(shift), Cn, Cs, eps, momentum, int(update), ptr(bwd_sums), _gref(groups), stream())


def bn_apply(x, y, scale, shift, rows, Cn, Cs, act, groups=None):
    _call("cpcsv_bn_apply", ptr(x), ptr(y), dcode(x), ptr(scale), ptr(shift), rows, Cn, Cs, act, _gref(groups), stream())


def bn_apply_partials(x, y, partials, ldstat, gamma, beta, rmean, rvar, stat_out, bwd_sums, rows, Cn, Cs, act, eps, momentum, groups):
    _call("cpcsv_bn_apply_partials", ptr(x), ptr(y), dcode(x), ptr(partials), ldstat, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
          ptr(stat_out), ptr(bwd_sums), rows, Cn, Cs, act, eps, momentum, _gref(groups), stream())


def bn_bwd_reduce(dy, x, mean, invstd, gamma, beta, sums, rows, Cn, Cs, act, groups=None):
    _call("cpcsv_bn_bwd_reduce", ptr(dy), ptr(x), dcode(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(sums),
          rows, Cn, Cs, act, _gref(groups), stream())


def bn_bwd_apply(dy, x, dx, mean, invstd, gamma, beta, sums, dgamma, dbeta, rows, Cn, Cs, act, accumulate=0, gw_out=None,
                 sigma=None, eps=0.0, groups=None):
    _call("cpcsv_bn_bwd_apply", ptr(dy), ptr(x), ptr(dx), dcode(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
          ptr(sums), ptr(dgamma), ptr(dbeta), rows, Cn, Cs, act, accumulate, ptr(gw_out), ptr(sigma), eps, _gref(groups), stream())


def colsum(x, out, rows, Cn, Cs):
    _call("cpcsv_colsum", ptr(x), dcode(x), ptr(out), rows, Cn, Cs, stream())


def concat_pad(srcs, dst, rows, ldd):
    ps = [ptr(t) for t in srcs] + [None] * (4 - len(srcs))
    ws = [t.shape[1] for t in srcs] + [0] * (4 - len(srcs))
    _call("cpcsv_concat_pad", ps[0], ws[0], ps[1], ws[1], ps[2], ws[2], ps[3], ws[3], len(srcs), ptr(dst), dcode(dst),
          rows, ldd, stream())


def act_bwd(dy, y, dz, act):
    _call("cpcsv_act_bwd", ptr(dy), ptr(y), ptr(dz), dcode(y), y.numel(), act, stream())


def gate_fwd(a, b, out):
    _call("cpcsv_gate_fwd", ptr(a), ptr(b), ptr(out), dcode(a), a.numel(), stream())


def gate_bwd(dout, a, b, da, db):
    _call("cpcsv_gate_bwd", ptr(dout), ptr(a), ptr(b), ptr(da), ptr(db), dcode(a), a.numel(), stream())


def planar_to_nhwc(src, dst, frames, T, sB, sT, sC, Cn, HW, Cs):
    _call("cpcsv_planar_to_nhwc", ptr(src), dcode(src), ptr(dst), dcode(dst), frames, T, sB, sT, sC, Cn, HW, Cs, stream())


def nhwc_to_planar(src, dst, frames, T, sB, sT, sC, Cn, HW, Cs):
    _call("cpcsv_nhwc_to_planar", ptr(src), dcode(src), ptr(dst), dcode(dst), frames, T, sB, sT, sC, Cn, HW, Cs, stream())


def ingest_u8(src, planar, nhwc, frames, T, sB, sT, sC, Cn, HW, Cs, mean, std):
    _call("cpcsv_ingest_u8", ptr(src), ptr(planar), ptr(nhwc), dcode(nhwc) if nhwc is not None else 0, frames, T, sB, sT, sC, Cn,
          HW, Cs, ptr(mean), ptr(std), stream())


def copy2d(src, lds, scol0, dst, ldd, dcol0, rows, cols, accumulate=0, fill=False):
    """fill: the whole dst row [0, ldd) is written, zeros outside the copied window (mode 2 of the C entry point)."""
    _call("cpcsv_copy2d", ptr(src), dcode(src), lds, scol0, ptr(dst), dcode(dst), ldd, dcol0, rows, cols,
          2 if fill else accumulate, stream())


def im2col(x, out, F, H, W, Cs, Cn, k, s, p, ld, adjoint=False):
    _call("cpcsv_im2col", ptr(x), ptr(out), dcode(x), F, H, W, Cs, Cn, k, s, p, ld, int(adjoint), stream())


def cond_concat(feat, cond, out, N, P, Cn, Cs_f, E, Cs_out):
    _call("cpcsv_cond_concat", ptr(feat), ptr(cond), ptr(out), dcode(feat), N, P, Cn, Cs_f, E, Cs_out, stream())


def cond_triplet(feat, cond, out, N, P, Cn, Cs_f, E, Cs_out):
    _call("cpcsv_cond_triplet", ptr(feat), ptr(cond), ptr(out), dcode(feat), N, P, Cn, Cs_f, E, Cs_out, stream())


def cond_triplet_bwd(dout, dfeat, N, P, Cn, Cs_f, Cs_out):
    _call("cpcsv_cond_triplet_bwd", ptr(dout), ptr(dfeat), dcode(dout), N, P, Cn, Cs_f, Cs_out, stream())


def bce_groups(p, t, out, grad, n0, n1, n2, w0, w1, w2):
    _call("cpcsv_bce_groups", ptr(p), ptr(t), ptr(out), ptr(grad), n0, n1, n2, float(w0), float(w1), float(w2), stream())


def mean_t(x, out, N, T, inner):
    _call("cpcsv_mean_t", ptr(x), ptr(out), dcode(x), N, T, inner, stream())


def mean_t_bwd(dout, din, N, T, inner):
    _call("cpcsv_mean_t_bwd", ptr(dout), ptr(din), dcode(dout), N, T, inner, stream())


def fill_zero(t):
    _call("cpcsv_fill_zero", ptr(t), t.numel() * t.element_size(), stream())


def gru_gates_fwd(gi, gh, h, hnew, gates, B, H, ldg, ldh):
    _call("cpcsv_gru_gates_fwd", ptr(gi), ptr(gh), ptr(h), ptr(hnew), ptr(gates), B, H, ldg, ldh, stream())


def gru_gates_bwd(dhnew, gates, h, dgi, dgh, dh, B, H, ldg, ldh):
    _call("cpcsv_gru_gates_bwd", ptr(dhnew), ptr(gates), ptr(h), ptr(dgi), ptr(dgh), ptr(dh), B, H, ldg, ldh, stream())


def dfl1d_fwd(sig, taps, out, N, Cn, Ln, K, pad):
    _call("cpcsv_dfl1d_fwd", ptr(sig), ptr(taps), ptr(out), N, Cn, Ln, K, pad, stream())


def dfl1d_bwd(dout, sig, taps, dsig, dtaps, N, Cn, Ln, K, pad):
    _call("cpcsv_dfl1d_bwd", ptr(dout), ptr(sig), ptr(taps), ptr(dsig), ptr(dtaps), N, Cn, Ln, K, pad, stream())


def reparam_fwd(mu, lv, eps, out):
    _call("cpcsv_reparam_fwd", ptr(mu), ptr(lv), ptr(eps), ptr(out), mu.numel(), stream())


def reparam_bwd(dout, lv, eps, dmu, dlv, accumulate=0):
    _call("cpcsv_reparam_bwd", ptr(dout), ptr(lv), ptr(eps), ptr(dmu), ptr(dlv), dout.numel(), accumulate, stream())


def bce_fwd(p, t, loss, grad):
    _call("cpcsv_bce_fwd", ptr(p), ptr(t), ptr(loss), ptr(grad), p.numel(), stream())


def mlsm_fwd(x, t, loss, grad, N, Cn, ld, acc=None):
    _call("cpcsv_mlsm_fwd", ptr(x), ptr(t), ptr(loss), ptr(grad), ptr(acc), N, Cn, ld, stream())


def scalar_list(xs, ws):
    l = L.ScalarList()
    l.n = len(xs)
    for i, (x, w) in enumerate(zip(xs, ws)):
        l.x[i], l.w[i] = x.data_ptr(), float(w)
    return l


def lincomb_fwd(lst, out):
    _call("cpcsv_lincomb_fwd", C.byref(lst), ptr(out), stream())


def lincomb_bwd(g, lst, dx):
    _call("cpcsv_lincomb_bwd", ptr(g), C.byref(lst), ptr(dx), stream())


def copy_many(pairs):
    """[(dst, src)] contiguous same-size tensors, any number: one launch per 8 pairs (cpcsv_copy_many)."""
    for i in range(0, len(pairs), 8):
        chunk = pairs[i:i + 8]
        l = L.CopyList()
        l.n = len(chunk)
        for k, (dst, src) in enumerate(chunk):
            l.dst[k], l.src[k], l.bytes[k] = dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size()
        _call("cpcsv_copy_many", C.byref(l), stream())


def pack_dense_many(jobs):
    """[(w [cout][cin] fp32, fwd or None, lin or None, cout, cin, cin_s, cout_s)]: one launch per 16 layers (cpcsv_pack_dense_many)."""
    for i in range(0, len(jobs), L.PACK_JOBS):
        chunk = jobs[i:i + L.PACK_JOBS]
        l = L.PackList()
        l.n = len(chunk)
        for k, (w, fwd, lin, cout, cin, cin_s, cout_s) in enumerate(chunk):
            j = l.j[k]
            j.w, j.fwd, j.lin = w.data_ptr(), (fwd.data_ptr() if fwd is not None else None), (lin.data_ptr() if lin is not None else None)
            j.cout, j.cin, j.cin_s, j.cout_s = cout, cin, cin_s, cout_s
        _call("cpcsv_pack_dense_many", C.byref(l), stream())


def kl_fwd(mu, lv, loss, dmu, dlv):
    _call("cpcsv_kl_fwd", ptr(mu), ptr(lv), ptr(loss), ptr(dmu), ptr(dlv), mu.numel(), stream())


def mse_fwd(a, b, loss, da, db, count=0):
    _call("cpcsv_mse_fwd", ptr(a), ptr(b), dcode(a), ptr(loss), ptr(da), ptr(db), a.numel(), int(count), stream())


def scale_by(x, y, alpha, mult=1.0, accumulate=0):
    _call("cpcsv_scale_by", ptr(x), ptr(y), dcode(x), ptr(alpha), float(mult), x.numel(), accumulate, stream())


def adam_step(table, sizes, ntensors, total_chunks, chunk_tensor, chunk_offset, hyper, b1, b2, eps):
    _call("cpcsv_adam_step", ptr(table), ptr(sizes), ntensors, total_chunks, ptr(chunk_tensor), ptr(chunk_offset),
          ptr(hyper), float(b1), float(b2), float(eps), stream())


def adam_chunk():
    return L.load().cpcsv_adam_chunk()


def logit_groups(rows, sigmas=None, us=None, vs=None):
    """cpcsv_logit_groups: cumulative row offsets [0, ..., R] + per-call spectral-norm state (tensors or None)."""
    g = L.LogitGroups()
    g.n = len(rows) - 1
    for i, r in enumerate(rows):
        g.row[i] = r
    for i in range(g.n):
        g.sigma[i] = ptr(sigmas[i]) if sigmas is not None else None
        g.u[i] = ptr(us[i]) if (us is not None and us[i] is not None) else None
        g.v[i] = ptr(vs[i]) if (vs is not None and vs[i] is not None) else None
    return g


def logit_head_fwd(x, w, bias, p, R, Kdim, groups):
    _call("cpcsv_logit_head_fwd", ptr(x), ptr(w), ptr(bias), ptr(p), dcode(x), R, Kdim, C.byref(groups), stream())


def logit_head_bwd(dy, p, w, dx, dz, dtype, R, Kdim, groups):
    _call("cpcsv_logit_head_bwd", ptr(dy), ptr(p), ptr(w), ptr(dx), ptr(dz), dtype, R, Kdim, C.byref(groups), stream())


def logit_head_scratch(Kdim, ngroups):
    return L.load().cpcsv_logit_head_scratch(Kdim, ngroups)


def logit_head_wgrad(dz, x, w, scratch, dW, db, R, Kdim, Cin, Cin_s, taps, groups):
    _call("cpcsv_logit_head_wgrad", ptr(dz), ptr(x), ptr(w), ptr(scratch), ptr(dW), ptr(db), dcode(x), R, Kdim, Cin, Cin_s, taps,
          C.byref(groups), stream())


def cond_head_max_samples():
    return L.load().cpcsv_cond_head_max_samples()


def cond_head_desc(counts, feat0, cond0):
    """Static part of a cpcsv_cond_head: the reference calls sharing the launch (samples per call, first feature sample, first condition row)."""
    d = L.CondHead()
    d.MH = d.MW = 4
    d.ngroups = len(counts)
    for g, (c, f0, c0) in enumerate(zip(counts, feat0, cond0)):
        d.count[g], d.feat0[g], d.cond0[g] = c, f0, c0
    return d


def cond_head_fwd(d, ws, nslabs, pt, alphas, z, y, gamma, beta, rmean, rvar, stat_out, pstride, bwd_sums, C_, act, eps, momentum):
    d.ws, d.nslabs, d.ldws, d.ws_rows = ws.data_ptr(), nslabs, ws.shape[-1], ws.shape[-2]
    d.pt, d.ldp = pt.data_ptr(), pt.shape[-1]
    for g in range(4):
        d.galpha[g] = ptr(alphas[g]) if (alphas is not None and g < len(alphas)) else None
    d.z, d.y, d.dtype, d.C, d.Cs = z.data_ptr(), y.data_ptr(), dcode(z), C_, z.shape[-1]
    d.gamma, d.beta, d.running_mean, d.running_var = ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar)
    d.stat_out, d.pstride, d.bwd_sums, d.act, d.eps, d.momentum = ptr(stat_out), pstride, int(bwd_sums), act, eps, momentum
    _call("cpcsv_cond_head_fwd", C.byref(d), stream())


def cond_head_bwd(counts, feat0, cond0, dz, dF, dZt, nfeat, ncond):
    d = L.CondHeadGrad()
    d.dz, d.dF, d.dZt, d.dtype = dz.data_ptr(), dF.data_ptr(), ptr(dZt), dcode(dz)
    d.MH = d.MW = 4
    d.Cs, d.nfeat, d.ncond, d.ngroups = dz.shape[-1], nfeat, ncond, len(counts)
    for g, (c, f0, c0) in enumerate(zip(counts, feat0, cond0)):
        d.count[g], d.feat0[g], d.cond0[g] = c, f0, c0
    _call("cpcsv_cond_head_bwd", C.byref(d), stream())


def batch_prep(im_desc, im_lab, im_cont, st_desc, st_lab, td):
    """cpcsv_batch_prep: -> (im_motion, im_content, st_motion, st_text, st_text_mean, chars), all contiguous fp32 (reference trainer.py:254-304)."""
    im, st, t, l = im_desc.shape[0], st_desc.shape[0], st_desc.shape[1], im_lab.shape[1]
    dev = im_desc.device
    mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    im_motion, im_content = mk(im, td + l), mk(im, t, td)
    st_motion, st_text, st_mean, chars = mk(st, t, td + l), mk(st, t, td), mk(st, td), mk(st, l)
    _call("cpcsv_batch_prep", ptr(im_desc), im_desc.stride(0), ptr(im_lab), ptr(im_cont), im_cont.stride(1), ptr(st_desc), st_desc.stride(1),
          ptr(st_lab), ptr(im_motion), ptr(im_content), ptr(st_motion), ptr(st_text), ptr(st_mean), ptr(chars), im, st, t, td, l, stream())
    return im_motion, im_content, st_motion, st_text, st_mean, chars
