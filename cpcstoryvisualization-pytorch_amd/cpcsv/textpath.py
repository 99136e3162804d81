"""The generator's text / motion encoders as ~10 stage launches forward and ~11 backward (csrc/text.hip, cpcsv_text_stage).

CA_NET, m_net, c_net, the two GRU recurrences, image_net, filter_net and DynamicFilterLayer1D (reference model.py:37-65,302-346,
371-378; layers.py:69-80) for BOTH calls of a generator pass (sample_videos on the stories, sample_images on the images:
reference trainer.py:295-300,367-369). Per layer they were 40-odd dependent 8-20 us launches at the head of every generator pass
and ~90 at the tail of its backward: the GPU idles through all of them. Here a STAGE is one launch that runs every job whose
inputs are ready - the two calls side by side, the motion chain beside the content chain - and a pass is the dependency depth of
the path: forward  prep | products without a recurrence (CA_NET, m_net+BN, both W_ih products, image_net+BN+tanh) | c_net+BN and
the GRU steps in lockstep | filter_net+BN | dynamic filter + concat;  backward the mirror image, then ONE weight-gradient launch
(cpcsv_dense_rows_wgrad_multi). Same arithmetic as the per-layer path (exact fp32 FMA chains, train-mode BatchNorm1d per call
with running statistics advanced story call first, the reference's noise draw order), another summation order.

`text_path(gen, ...)` is what model.StoryGAN.sample_both calls; `supported(...)` says when (CPCSV_TEXT_FUSED=0: never).
"""
import ctypes as C
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import kernels as K
from . import runtime as R
from .runtime import pad8, stream, tdtype

ENABLED = os.environ.get("CPCSV_TEXT_FUSED", "1") != "0"


def _p(t, off=0):
    """device address of element `off` of tensor t (None stays None)"""
    if t is None:
        return None
    return t.data_ptr() + off * t.element_size()


class _Stages:
    """stage index -> cpcsv_txt_stage under construction"""

    def __init__(self):
        self.st = {}

    def add(self, i, typ, M, T=(0, 0), N=0, Kd=0, ldx=0, ldw=0, ldy=0, act=0, A=(), eps=0.0, mom=0.0, x=(), w=None, bias=None, y=(),
            P=((), ()), Q=()):
        s = self.st.get(i)
        if s is None:
            s = self.st[i] = L.TxtStage()
        if s.njobs >= L.TXT_MAX_JOBS:
            raise RuntimeError("text stage %d holds more than %d jobs" % (i, L.TXT_MAX_JOBS))
        j = s.job[s.njobs]
        s.njobs += 1
        j.type, j.npass = typ, len(M)
        for p in range(len(M)):
            j.M[p] = int(M[p])
            j.T[p] = int(T[p]) if p < len(T) else 0
            j.x[p] = x[p] if p < len(x) else None
            j.y[p] = y[p] if p < len(y) else None
            row = P[p] if p < len(P) else ()
            for k in range(6):
                j.P[p][k] = row[k] if k < len(row) else None
        j.N, j.K, j.ldx, j.ldw, j.ldy, j.act = int(N), int(Kd), int(ldx), int(ldw), int(ldy), int(act)
        for k in range(8):
            j.A[k] = int(A[k]) if k < len(A) else 0
        j.eps, j.momentum = float(eps), float(mom)
        j.w, j.bias = w, bias
        for k in range(4):
            j.Q[k] = Q[k] if k < len(Q) else None

    def run(self):
        s_ = stream()
        for i in sorted(self.st):
            K._call("cpcsv_text_stage", C.byref(self.st[i]), s_)


class _Dims:
    def __init__(self, gen, bs, ts, bi):
        self.B, self.T = (bs, bi), (ts, 1)
        self.N = (bs * ts, bi)
        self.md, self.C, self.nz = gen.motion_dim, gen.content_dim, gen.noise_dim
        self.Lw, self.KF, self.nch = gen.image_size, gen.filter_size, gen.filter_num
        self.md_s, self.C_s, self.e_s = pad8(self.md), pad8(self.C), pad8(self.nz + self.md)
        self.g_m, self.g_c = pad8(3 * self.md), pad8(3 * self.C)
        self.i_n, self.f_n = self.nch * self.Lw, self.nch * self.KF
        self.i_s, self.f_s = pad8(self.i_n), pad8(self.f_n)
        self.tdim = gen.ca_net.t_dim
        self.zw = self.md + self.C + self.Lw
        self.z_s = pad8(self.zw)


def _layers(gen):
    """the nine small dense KernelLayers (operand copies) and the four BatchNorm1d holders, cached on the generator"""
    lay = gen.__dict__.get("_txt_layers")
    if lay is None:
        from . import modules as M
        ih_m, hh_m = gen.recurrent._layers()
        ih_c, hh_c = gen.mocornn._layers()
        lay = gen.__dict__["_txt_layers"] = {
            "ca": M._layer_for(gen.ca_net.fc, None, L.ACT_RELU, 0, out_mode="f32pad"),
            "m": gen.m_net._plan()[0], "c": gen.c_net._plan()[0], "i": gen.image_net._plan()[0], "f": gen.filter_net._plan()[0],
            "ih_m": ih_m, "hh_m": hh_m, "ih_c": ih_c, "hh_c": hh_c}
    else:
        gen.recurrent._layers()       # (re-points the GRU holders at the current parameters)
        gen.mocornn._layers()
    return lay


def supported(gen, st_motion, st_content, im_motion, im_content):
    """The fused path serves a training-mode generator on the GPU whose calls fit one tile (<= 64 frame rows per call: the weight
    gradient launch walks <= 64-row pieces) with 16-byte aligned text rows; everything else takes the per-layer path."""
    if not ENABLED or not st_motion.is_cuda or not gen.training or st_motion.dim() != 3 or im_motion.dim() != 2:
        return False
    bs, ts, bi = st_motion.shape[0], st_motion.shape[1], im_motion.shape[0]
    if ts < 2 or ts != gen.video_len or bs * ts > 64 or bi > 64 or bs < 2 or bi < 2:
        return False
    if gen.ca_net.t_dim % 4 or gen.out_num != 1 or st_motion.shape[2] != gen.motion_dim or im_motion.shape[1] != gen.motion_dim:
        return False
    if st_content.shape[0] != bs or im_content.shape[0] != bi:
        return False
    d = _Dims(gen, bs, ts, bi)
    if d.Lw + d.nch * d.Lw + d.nch * d.KF > 1024:
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in (gen.ca_net.fc.weight, gen.recurrent.weight_hh))


def _params(gen):
    m, c, i, f = gen.m_net, gen.c_net, gen.image_net, gen.filter_net
    return (gen.ca_net.fc.weight, gen.ca_net.fc.bias,
            m[0].weight, m[0].bias, m[1].weight, m[1].bias,
            c[0].weight, c[0].bias, c[1].weight, c[1].bias,
            gen.recurrent.weight_ih, gen.recurrent.bias_ih, gen.recurrent.weight_hh, gen.recurrent.bias_hh,
            gen.mocornn.weight_ih, gen.mocornn.bias_ih, gen.mocornn.weight_hh, gen.mocornn.bias_hh,
            i[0].weight, i[0].bias, i[1].weight, i[1].bias,
            f[0].weight, f[0].bias, f[1].weight, f[1].bias)


_WNAMES = ("ca", "m", "c", "ih_m", "hh_m", "ih_c", "hh_c", "i", "f")
_WIDX = {"ca": 0, "m": 2, "c": 6, "ih_m": 10, "hh_m": 12, "ih_c": 14, "hh_c": 16, "i": 18, "f": 22}     # weight index in _params (bias = +1)
_BNIDX = {"m": 4, "c": 8, "i": 20, "f": 24}                                                         # gamma index in _params (beta = +1)


def text_path(gen, st_motion, st_flat, im_motion, im_flat, draw_ca, draw):
    """-> (zmc [B_s*T + B_i, pad8(md + C + L)] in the compute dtype, r_mu, r_logvar, c_mu, c_logvar). `draw_ca(shape)` / `draw(shape)`
    hand out the N(0,1) noise of CA_NET / of the generator in the reference's order (model.py:56-58,315,319)."""
    bs, ts, bi = st_motion.shape[0], st_motion.shape[1], im_motion.shape[0]
    d = _Dims(gen, bs, ts, bi)
    noise = []
    for b, t in ((bs, ts), (bi, 1)):
        eps = draw_ca((b, d.C))                      # CA_NET.reparametrize (drawn for the image call too: its code is unused)
        n0 = draw((b, d.md))                         # get_gru_initial_state
        if gen.noise_source is None:
            z = draw((t * b, d.nz))                  # the T step noises as one time-major draw
        else:
            zs = [draw((b, d.nz)) for _ in range(t)]
            z = zs[0] if t == 1 else torch.cat(zs, 0)
        noise += [eps.contiguous(), n0.contiguous(), z.contiguous()]
    return TextPathFn.apply(gen, st_motion.contiguous(), st_flat.contiguous(), im_motion.contiguous(), im_flat.contiguous(),
                            *noise, *_params(gen))


class TextPathFn(Function):
    @staticmethod
    def forward(ctx, gen, st_motion, st_flat, im_motion, im_flat, eps_s, n0_s, z_s, eps_i, n0_i, z_i, *params):
        R.require_gpu(st_motion)
        dev = st_motion.device
        bs, ts, bi = st_motion.shape[0], st_motion.shape[1], im_motion.shape[0]
        d = _Dims(gen, bs, ts, bi)
        lay = _layers(gen)
        grad = any(ctx.needs_input_grad[11:])
        packs = {}
        for nm in _WNAMES:
            w = params[_WIDX[nm]]
            packs[nm] = lay[nm].packs(w, L.F32, "both" if grad else "fwd")            # (rebuilt by model._prepack_text in one launch; cached here)
        bns = {"m": gen.m_net[1], "c": gen.c_net[1], "i": gen.image_net[1], "f": gen.filter_net[1]}
        f32 = torch.float32
        B, T, N = d.B, d.T, d.N
        # ---- workspace: one allocation, carved into the per-call buffers (kept for the backward pass) ----
        spec = []
        for p in range(2):
            spec += [("mpad", p, N[p] * d.md_s), ("tpad", p, N[p] * d.md_s), ("e", p, N[p] * d.e_s), ("n0pad", p, B[p] * d.md_s),
                     ("crnn", p, N[p] * d.C_s),
                     ("xca", p, B[p] * 2 * d.C), ("mu", p, B[p] * d.C), ("lv", p, B[p] * d.C), ("code", p, B[p] * d.C_s),
                     ("lin_m", p, B[p] * d.md_s), ("save_m", p, 2 * d.md_s), ("hall_m", p, (T[p] + 1) * B[p] * d.md_s),
                     ("gi_m", p, N[p] * d.g_m), ("gates_m", p, N[p] * 4 * d.md),
                     ("lin_c", p, B[p] * d.C_s), ("save_c", p, 2 * d.C_s), ("hall_c", p, (T[p] + 1) * B[p] * d.C_s),
                     ("gi_c", p, N[p] * d.g_c), ("gates_c", p, N[p] * 4 * d.C),
                     ("lin_i", p, N[p] * d.i_s), ("save_i", p, 2 * d.i_s), ("mimg", p, N[p] * d.i_s),
                     ("lin_f", p, N[p] * d.f_s), ("save_f", p, 2 * d.f_s), ("cfilt", p, N[p] * d.f_s)]
        off, total = {}, 0
        for nm, p, n in spec:
            off[(nm, p)] = total
            total += (n + 3) // 4 * 4                                                  # every buffer 16-byte aligned
        ws = torch.empty(total, dtype=f32, device=dev)
        if os.environ.get("CPCSV_POISON", "0") == "1":
            ws.fill_(float("nan"))
        a = lambda nm, p, o=0: ws.data_ptr() + 4 * (off[(nm, p)] + o)
        out_dt = tdtype()
        zmc = torch.empty(N[0] + N[1], d.z_s, dtype=out_dt, device=dev)
        obf = 1 if out_dt == torch.bfloat16 else 0
        P = _p
        prm = lambda i: P(params[i])
        fwd = lambda nm: P(packs[nm][0])
        motion = (st_motion, im_motion)
        text = (st_flat, im_flat)
        eps = (eps_s, None)                                                            # sample_images feeds the MEAN to c_net (model.py:433)
        n0, zn = (n0_s, n0_i), (z_s, z_i)
        bn = lambda k: (prm(_BNIDX[k]), prm(_BNIDX[k] + 1), P(bns[k].running_mean), P(bns[k].running_var))
        S = _Stages()
        both = (0, 1)
        # stage 0: padded operand matrices (time-major rows)
        S.add(0, L.TXT_PREP, B, T, A=(d.md, d.nz, d.md_s, d.e_s), y=[a("tpad", p) for p in both],
              P=[(P(motion[p]), P(zn[p]), P(n0[p]), a("mpad", p), a("e", p), a("n0pad", p)) for p in both])
        # stage 1: every product that waits for no recurrence
        S.add(1, L.TXT_CA, B, Kd=d.tdim, ldx=d.tdim, ldw=lay["ca"].cin_s, A=(d.C, d.C_s), x=[P(text[p]) for p in both], w=fwd("ca"),
              bias=prm(1), y=[a("xca", p) for p in both], P=[(a("mu", p), a("lv", p), P(eps[p]), a("code", p)) for p in both])
        S.add(1, L.TXT_DENSE, B, N=d.md, Kd=d.md_s, ldx=d.md_s, ldw=lay["m"].cin_s, ldy=d.md_s, eps=bns["m"].eps, mom=bns["m"].momentum,
              x=[a("n0pad", p) for p in both], w=fwd("m"), bias=prm(3), y=[a("hall_m", p) for p in both],
              P=[(a("lin_m", p), a("save_m", p)) for p in both], Q=bn("m"))
        S.add(1, L.TXT_DENSE, N, N=3 * d.md, Kd=d.e_s, ldx=d.e_s, ldw=lay["ih_m"].cin_s, ldy=d.g_m, x=[a("e", p) for p in both],
              w=fwd("ih_m"), bias=prm(11), y=[a("gi_m", p) for p in both])
        S.add(1, L.TXT_DENSE, N, N=3 * d.C, Kd=d.md_s, ldx=d.md_s, ldw=lay["ih_c"].cin_s, ldy=d.g_c, x=[a("mpad", p) for p in both],
              w=fwd("ih_c"), bias=prm(15), y=[a("gi_c", p) for p in both])
        S.add(1, L.TXT_DENSE, N, N=d.i_n, Kd=d.md_s, ldx=d.md_s, ldw=lay["i"].cin_s, ldy=d.i_s, act=L.ACT_TANH, eps=bns["i"].eps,
              mom=bns["i"].momentum, x=[a("tpad", p) for p in both], w=fwd("i"), bias=prm(19), y=[a("mimg", p) for p in both],
              P=[(a("lin_i", p), a("save_i", p)) for p in both], Q=bn("i"))
        # stage 2: c_net + BatchNorm on the code; the recurrences in lockstep from here (motion step t at stage 2 + t, content step t
        # at stage 3 + t)
        S.add(2, L.TXT_DENSE, B, N=d.C, Kd=d.C_s, ldx=d.C_s, ldw=lay["c"].cin_s, ldy=d.C_s, eps=bns["c"].eps, mom=bns["c"].momentum,
              x=[a("code", p) for p in both], w=fwd("c"), bias=prm(7), y=[a("hall_c", p) for p in both],
              P=[(a("lin_c", p), a("save_c", p)) for p in both], Q=bn("c"))
        for t in range(T[0]):
            live = [p for p in both if t < T[p]]
            S.add(2 + t, L.TXT_GRU_FWD, [B[p] for p in live], Kd=d.md_s, ldx=d.md_s, ldw=lay["hh_m"].cin_s, ldy=d.md_s, A=(d.md, d.g_m),
                  x=[a("hall_m", p, t * B[p] * d.md_s) for p in live], w=fwd("hh_m"), bias=prm(13),
                  y=[a("hall_m", p, (t + 1) * B[p] * d.md_s) for p in live],
                  P=[(a("gi_m", p, t * B[p] * d.g_m), a("gates_m", p, t * B[p] * 4 * d.md)) for p in live])
            S.add(3 + t, L.TXT_GRU_FWD, [B[p] for p in live], [T[p] for p in live], Kd=d.C_s, ldx=d.C_s, ldw=lay["hh_c"].cin_s, ldy=d.C_s,
                  A=(d.C, d.g_c, t), x=[a("hall_c", p, t * B[p] * d.C_s) for p in live], w=fwd("hh_c"), bias=prm(17),
                  y=[a("hall_c", p, (t + 1) * B[p] * d.C_s) for p in live],
                  P=[(a("gi_c", p, t * B[p] * d.g_c), a("gates_c", p, t * B[p] * 4 * d.C), a("crnn", p)) for p in live])
        sf = 3 + T[0]
        S.add(sf, L.TXT_DENSE, N, N=d.f_n, Kd=d.C_s, ldx=d.C_s, ldw=lay["f"].cin_s, ldy=d.f_s, eps=bns["f"].eps, mom=bns["f"].momentum,
              x=[a("crnn", p) for p in both], w=fwd("f"), bias=prm(23), y=[a("cfilt", p) for p in both],
              P=[(a("lin_f", p), a("save_f", p)) for p in both], Q=bn("f"))
        S.add(sf + 1, L.TXT_JOINT, B, T, ldx=d.i_s, ldw=d.f_s, ldy=d.z_s, A=(d.md, d.C, d.Lw, d.KF, d.nch, d.md_s, obf),
              y=[P(zmc), P(zmc, N[0] * d.z_s)], P=[(a("hall_m", p), a("mu", p), a("mimg", p), a("cfilt", p)) for p in both])
        S.run()
        for b_ in bns.values():                                                        # two train-mode calls per BatchNorm layer
            b_.note_batch()
            b_.note_batch()
        view = lambda nm, p, *shape: ws[off[(nm, p)]:off[(nm, p)] + int(torch.Size(shape).numel())].view(*shape)
        outs = (zmc, view("mu", 0, B[0], d.C), view("lv", 0, B[0], d.C), view("mu", 1, B[1], d.C), view("lv", 1, B[1], d.C))
        ctx.gen, ctx.d, ctx.off, ctx.lay = gen, d, off, lay
        ctx.save_for_backward(ws, st_flat, im_flat, eps_s, *params)
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dmu_s, dlv_s, dmu_i, dlv_i):
        gen, d, off, lay = ctx.gen, ctx.d, ctx.off, ctx.lay
        saved = ctx.saved_tensors
        ws, st_flat, im_flat, eps_s = saved[:4]
        params = saved[4:]
        dev = ws.device
        f32 = torch.float32
        B, T, N = d.B, d.T, d.N
        both = (0, 1)
        P = _p
        prm = lambda i: P(params[i])
        a = lambda nm, p, o=0: ws.data_ptr() + 4 * (off[(nm, p)] + o)
        packs = {nm: lay[nm].packs(params[_WIDX[nm]], L.F32, "bwd") for nm in ("hh_m", "hh_c", "c", "f")}
        lin = lambda nm: P(packs[nm][2])
        cont = lambda t: None if t is None else t.contiguous().float()
        dz = dz.contiguous()
        ibf = 1 if dz.dtype == torch.bfloat16 else 0
        dmu_ext, dlv_ext = (cont(dmu_s), cont(dmu_i)), (cont(dlv_s), cont(dlv_i))
        # ---- gradient workspace ----
        spec = []
        for p in both:
            spec += [("dpre", p, N[p] * d.i_s), ("dflt", p, N[p] * d.f_s), ("dhe_m", p, N[p] * d.md_s), ("dmu", p, B[p] * d.C),
                     ("dlin_i", p, N[p] * d.i_s), ("dlin_f", p, N[p] * d.f_s), ("dcrnn", p, N[p] * d.C_s),
                     ("dgi_m", p, N[p] * d.g_m), ("dgh_m", p, N[p] * d.g_m), ("dhz_m", p, 2 * B[p] * d.md_s),
                     ("dgi_c", p, N[p] * d.g_c), ("dgh_c", p, N[p] * d.g_c), ("dhz_c", p, 2 * B[p] * d.C_s),
                     ("dlin_m", p, B[p] * d.md_s), ("dlin_c", p, B[p] * d.C_s), ("dxca", p, B[p] * 2 * d.C)]
        goff, total = {}, 0
        for nm, p, n in spec:
            goff[(nm, p)] = total
            total += (n + 3) // 4 * 4
        gw = torch.empty(total, dtype=f32, device=dev)
        if os.environ.get("CPCSV_POISON", "0") == "1":
            gw.fill_(float("nan"))
        g = lambda nm, p, o=0: gw.data_ptr() + 4 * (goff[(nm, p)] + o)
        # ---- where the parameter gradients go: the persistent flat gradient buffer (accumulated in place, autograd sees None) or
        # a fresh zero tensor handed back to autograd
        direct = lambda q: getattr(q, "_cpcsv_direct", False) and q.grad is not None and q.grad.is_contiguous() \
            and not getattr(q, "_cpcsv_retired", False)
        grads = [None] * len(params)
        tgt = []
        for i, q in enumerate(params):
            if not ctx.needs_input_grad[11 + i]:
                tgt.append(None)
            elif direct(q):
                tgt.append(q.grad)
            else:
                grads[i] = torch.zeros_like(q)
                tgt.append(grads[i])
        bnq = lambda k: (prm(_BNIDX[k]), P(tgt[_BNIDX[k]]), P(tgt[_BNIDX[k] + 1]))
        S = _Stages()
        # stage 0: dynamic filter backward (+ tanh'), the motion states' direct gradients, the gradients of mu
        S.add(0, L.TXT_DFL_BWD, B, T, ldx=d.z_s, ldw=d.i_s, ldy=d.f_s, A=(d.md, d.C, d.Lw, d.KF, d.nch, d.md_s, ibf),
              x=[P(dz), P(dz, N[0] * d.z_s)], y=[P(dmu_ext[p]) for p in both],
              P=[(a("mimg", p), a("cfilt", p), g("dpre", p), g("dflt", p), g("dhe_m", p), g("dmu", p)) for p in both])
        # stage 1: BatchNorm backward of image_net / filter_net (two calls each, in call order inside the blocks)
        S.add(1, L.TXT_BN_BWD, N, N=d.i_n, ldx=d.i_s, ldy=d.i_s, x=[g("dpre", p) for p in both], y=[g("dlin_i", p) for p in both],
              P=[(a("lin_i", p), a("save_i", p), None) for p in both], Q=bnq("i"))
        S.add(1, L.TXT_BN_BWD, N, N=d.f_n, ldx=d.f_s, ldy=d.f_s, x=[g("dflt", p) for p in both], y=[g("dlin_f", p) for p in both],
              P=[(a("lin_f", p), a("save_f", p), None) for p in both], Q=bnq("f"))
        # stage 2: d crnn = d lin_f W_f (story-major rows, like crnn)
        S.add(2, L.TXT_DENSE, N, N=d.C, Kd=d.f_s, ldx=d.f_s, ldw=packs["f"][2].shape[1], ldy=d.C_s, x=[g("dlin_f", p) for p in both],
              w=lin("f"), y=[g("dcrnn", p) for p in both])

        def chain(p, first, tag, H, ld, gld, ext, ext_rows, lin_w, ldw, gamma_k, lin_key, save_key, dlin_key):
            """one call's recurrence backward: step t at stage first + (T - 1 - t), then dh_0 + BatchNorm backward"""
            Tp, Bp = T[p], B[p]
            for t in range(Tp - 1, -1, -1):
                st = first + (Tp - 1 - t)
                more = t < Tp - 1
                S.add(st, L.TXT_GRU_BWD, [Bp], Kd=gld if more else 0, ldx=gld, ldw=ldw, ldy=ld, A=(H, gld, ext_rows),
                      x=[g("dgh_" + tag, p, (t + 1) * Bp * gld)] if more else [None], w=lin_w if more else None,
                      y=[g("dhz_" + tag, p, (t % 2) * Bp * ld)],
                      P=[(ext(p, t), g("dhz_" + tag, p, ((t + 1) % 2) * Bp * ld) if more else None, a("gates_" + tag, p, t * Bp * 4 * H),
                          a("hall_" + tag, p, t * Bp * ld), g("dgi_" + tag, p, t * Bp * gld), g("dgh_" + tag, p, t * Bp * gld))])
            S.add(first + Tp, L.TXT_BN_BWD, [Bp], N=H, Kd=gld, ldx=gld, ldw=ldw, ldy=ld, x=[g("dgh_" + tag, p)], w=lin_w,
                  y=[g(dlin_key, p)], P=[(a(lin_key, p), a(save_key, p), g("dhz_" + tag, p))], Q=bnq(gamma_k))
            return first + Tp

        ldw_m, ldw_c = packs["hh_m"][2].shape[1], packs["hh_c"][2].shape[1]
        for p in both:
            chain(p, 1, "m", d.md, d.md_s, d.g_m, lambda p_, t: g("dhe_m", p_, t * B[p_] * d.md_s), 1, lin("hh_m"), ldw_m, "m", "lin_m",
                  "save_m", "dlin_m")
            # (d crnn is story-major like crnn itself: the rows of step t are every T-th one from row t)
            end_c = chain(p, 3, "c", d.C, d.C_s, d.g_c, lambda p_, t: g("dcrnn", p_, t * d.C_s), T[p], lin("hh_c"), ldw_c, "c", "lin_c",
                          "save_c", "dlin_c")
            # d code -> d (CA_NET output before the split), through the reparametrisation (story call) and the ReLU
            S.add(end_c + 1, L.TXT_CA_BWD, [B[p]], Kd=d.C_s, ldx=d.C_s, ldw=packs["c"][2].shape[1], A=(d.C,), x=[g("dlin_c", p)], w=lin("c"),
                  y=[g("dxca", p)], P=[(g("dmu", p), P(dlv_ext[p]), P(eps_s) if p == 0 else None, a("xca", p))])
        S.run()
        # ---- all weight / bias gradients: ONE launch (cpcsv_dense_rows_wgrad_multi; parked when the enclosing backward batches them)
        wsv = lambda nm, p, rows, ld, o=0: ws[off[(nm, p)] + o:off[(nm, p)] + o + rows * ld].view(rows, ld)
        gv = lambda nm, p, rows, ld: gw[goff[(nm, p)]:goff[(nm, p)] + rows * ld].view(rows, ld)
        text = (st_flat, im_flat)
        pieces = []
        for p in both:
            Bp, Np, Tp = B[p], N[p], T[p]
            pieces += [
                ("ca", gv("dxca", p, Bp, 2 * d.C), text[p], Bp, 2 * d.C, d.tdim),
                ("c", gv("dlin_c", p, Bp, d.C_s), wsv("code", p, Bp, d.C_s), Bp, d.C, d.C),
                ("m", gv("dlin_m", p, Bp, d.md_s), wsv("n0pad", p, Bp, d.md_s), Bp, d.md, d.md),
                ("ih_m", gv("dgi_m", p, Np, d.g_m), wsv("e", p, Np, d.e_s), Np, 3 * d.md, d.nz + d.md),
                ("hh_m", gv("dgh_m", p, Np, d.g_m), wsv("hall_m", p, Np, d.md_s), Np, 3 * d.md, d.md),
                ("ih_c", gv("dgi_c", p, Np, d.g_c), wsv("mpad", p, Np, d.md_s), Np, 3 * d.C, d.md),
                ("hh_c", gv("dgh_c", p, Np, d.g_c), wsv("hall_c", p, Np, d.C_s), Np, 3 * d.C, d.C),
                ("i", gv("dlin_i", p, Np, d.i_s), wsv("tpad", p, Np, d.md_s), Np, d.i_n, d.md),
                ("f", gv("dlin_f", p, Np, d.f_s), wsv("crnn", p, Np, d.C_s), Np, d.f_n, d.C)]
        deferred = R.small_wgrads_deferred()
        if not deferred:
            R.discard_small_wgrads()           # no deferred region is open: whatever an aborted earlier backward left parked is stale
        parked = 0
        for nm, dzt, xt, rows, n, kr in pieces:
            wi = _WIDX[nm]
            if tgt[wi] is None:
                continue
            R.park_small_wgrad(tgt[wi], tgt[wi + 1], dzt, xt, rows, n, kr)
            parked += 1
        if parked and not deferred:
            R.flush_small_wgrads()             # exactly this call's pieces
        return (None,) * 11 + tuple(grads)
