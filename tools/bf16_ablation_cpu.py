#!/usr/bin/env python3
"""Which bf16-stored tensor carries the generator-gradient error of the bf16 training step?

CPU experiment on the oracle (test infrastructure): the fp64 oracle step at cfg/final.yml widths is the yardstick; the
same fp64 step is then re-run with bf16 ROUNDING emulated at one class of storage points at a time - exactly the
tensors the HIP path keeps in bf16:
    W   packed conv / big-dense weights (operands of the MFMAs)
    Z   pre-BatchNorm conv outputs (y_raw, re-read by the BN backward)
    Y   post-activation layer outputs (the next layer's input), incl. the fakes entering the critics
    DZ  gradient w.r.t. the conv output (dz, operand of dgrad / wgrad GEMMs)
    DY  gradient w.r.t. the layer output (dx of the next layer's dgrad GEMM)
Small dense layers (cin*cout <= 2^21: text encoders, GRU) compute in fp32 in the product and are left alone.
Reports relative L2 error and cosine of the generator's / critics' parameter gradients against the un-rounded fp64 run.

    python tools/bf16_ablation_cpu.py <st> <im> <cascade 0|1> [variant ...]      (variants: names joined by '+')
"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.cpcsv_oracle import NoiseTape, make_state, pororo_cfg, synthetic_batch, train_step
from oracle.cpcsv_oracle import nets as N


class RoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None, None


def rnd(x, fwd, bwd):
    if not (fwd or bwd) or x is None:
        return x
    return RoundFn.apply(x, bool(fwd), bool(bwd))


FLAGS = set()
SCOPE = {"G", "D"}          # which nets get the emulation
ONLY = [None]               # module-name prefixes (within a net) the emulation is restricted to, or None


def big(m):
    if isinstance(m, nn.Linear):
        return m.in_features * m.out_features > (1 << 21)
    return True


def patch(net, tag):
    """Wrap the conv / big dense layers and the activations that follow them."""
    on_net = lambda f: (f in FLAGS) and (tag in SCOPE)
    for name, m in net.named_modules():
        hit = ONLY[0] is None or any(name == p or name.startswith(p + ".") for p in ONLY[0])
        on = (lambda f, hit=hit: on_net(f) and hit)
        if isinstance(m, (nn.Conv2d, nn.Linear)) and big(m):
            def fwd(x, m=m, on=on):
                w = rnd(m.weight, on("W"), False)
                x = rnd(x, on("Y"), False)       # operand as stored (already rounded when it is a layer output)
                if isinstance(m, nn.Conv2d):
                    z = F.conv2d(x, w, m.bias, m.stride, m.padding)
                else:
                    z = F.linear(x, w, m.bias)
                return rnd(z, on("Z"), on("DZ"))
            m.forward = fwd
        elif isinstance(m, (N.SpectralConv2d,)):
            def fwd(x, m=m, on=on):
                # the product packs W_orig in bf16 and applies 1/sigma (from the fp32 master) in the epilogue
                w = m.weight_orig
                wm = w.reshape(w.shape[0], -1)
                if m.training:
                    with torch.no_grad():
                        v = F.normalize(torch.mv(wm.t(), m.weight_u), dim=0, eps=1e-12)
                        u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
                        m.weight_v.copy_(v)
                        m.weight_u.copy_(u)
                u, v = m.weight_u.clone(), m.weight_v.clone()
                sigma = torch.dot(u, torch.mv(wm, v))
                x = rnd(x, on("Y"), False)
                z = F.conv2d(x, rnd(w, on("W"), False) / sigma, m.bias, m.stride, m.pad)
                return rnd(z, on("Z"), on("DZ"))
            m.forward = fwd
        elif isinstance(m, (nn.ReLU, nn.LeakyReLU, nn.Tanh)):
            parent = net.get_submodule(name.rsplit(".", 1)[0]) if "." in name else net
            sib = list(parent.children())
            if not any(isinstance(s, (nn.Conv2d, N.SpectralConv2d)) or (isinstance(s, nn.Linear) and big(s)) for s in sib):
                continue
            orig = m.forward

            def fwd(x, orig=orig, on=on):
                return rnd(orig(x), on("Y"), on("DY"))
            m.forward = fwd


def grads_err(got, want):
    num = den = dot = gg = 0.0
    for k, g in want.items():
        a = got[k].double()
        d = a - g.double()
        num += float((d * d).sum())
        den += float((g.double() ** 2).sum())
        dot += float((a * g.double()).sum())
        gg += float((a * a).sum())
    return (num / max(den, 1e-300)) ** 0.5, dot / max((den * gg) ** 0.5, 1e-300)


def main():
    st, im, casc = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
    variants = sys.argv[4:] or ["W", "Z", "Y", "DZ", "DY", "W+Z+Y", "DZ+DY", "W+Z+Y+DZ+DY", "Z+DZ", "W+Y+DY"]
    torch.set_default_dtype(torch.float64)
    oc = pororo_cfg(st_batch=st, im_batch=im, cascade=casc)
    state0 = make_state(oc, seed=0)
    names = ("G", "D_im", "D_st", "D_se")
    nets = lambda s_: (s_.netG, s_.netD_im, s_.netD_st, s_.netD_se)
    sds = {k: copy.deepcopy(n.state_dict()) for k, n in zip(names, nets(state0))}
    stb, imb = synthetic_batch(oc, seed=1)
    d = lambda b: {k: v.double() for k, v in b.items()}
    stb, imb = d(stb), d(imb)
    torch.manual_seed(5)
    t0 = time.time()
    ref = train_step(state0, stb, imb, noise=NoiseTape())
    tape = ref["noise_tape"]
    print("# fp64 oracle step ST=%d IM=%d cascade=%d: %.0f s" % (st, im, casc, time.time() - t0), flush=True)
    print("%-22s %-6s %10s %8s   %10s %10s %10s   %s" % ("rounded", "nets", "gradl2_G", "cos_G", "D_im", "D_st", "D_se", "G_loss rel"), flush=True)
    for var in variants:
        scope = {"G", "D"}
        v = var
        ONLY[0] = None
        if "@" in v:                         # W+Z+Y:G@upsample4,img  -> only those sub-modules
            v, only = v.split("@")
            ONLY[0] = only.split(",")
            var = v
        if ":" in v:
            v, sc = v.split(":")
            scope = set(sc.split(","))
        FLAGS.clear()
        FLAGS.update(v.split("+"))
        SCOPE.clear()
        SCOPE.update(scope)
        s = make_state(oc, seed=0)
        for k, n in zip(names, nets(s)):
            n.load_state_dict(sds[k])
            patch(n, "G" if k == "G" else "D")
        t0 = time.time()
        out = train_step(s, stb, imb, noise=NoiseTape(tape))
        eg, cg = grads_err(out["grads_G"], ref["grads_G"])
        ed = [grads_err(out["grads_" + k], ref["grads_" + k])[0] for k in ("D_im", "D_st", "D_se")]
        print("%-22s %-6s %10.4f %8.4f   %10.4f %10.4f %10.4f   %.2e   (%.0f s)" % (
            v + ("@" + ",".join(ONLY[0]) if ONLY[0] else ""), ",".join(sorted(scope)), eg, cg, ed[0], ed[1], ed[2], abs(out["G_loss"] - ref["G_loss"]) / abs(ref["G_loss"]),
            time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
