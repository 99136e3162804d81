#!/usr/bin/env python3
"""What bf16 costs on a TRAINED state (the random-init figures of tests/test_gpu_fullsize.py are the other end):
   (1) TRAIN_STEPS product steps in fp32 at cfg/final.yml widths, ST/IM as given (default the bench batch 12/60), rotating
       synthetic batches;  (2) snapshot of every net's state_dict;  (3) ONE step from that state by the fp64 oracle (CPU),
       recording its noise;  (4) the same step - same weights, batch, noise - by the product in bf16 and in fp32, and by the
       fp32 oracle: relative L2 / cosine / length ratio of every net's whole gradient vector against fp64, and the losses.
   python tools/bf16_trained_state.py [--cascade] [--steps 300] [--st 12]  >> profiles/r04_bf16_trained_state.txt"""
import argparse
import copy
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")):
    sys.path.insert(0, p)
if "torch" not in sys.modules:
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cascade", action="store_true")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--st", type=int, default=12)
    ap.add_argument("--detail", action="store_true", help="per-tensor split of the generator's gradient error; the fp32 oracle re-run "
                    "with BatchNorm statistics computed the way the product computes them (fp32 tile partials, single pass)")
    ap.add_argument("--arms", default="fp32,bf16")
    ap.add_argument("--frozen", action="store_true", help="also compare with the critics' learning rate set to 0 (see evaluate)")
    ap.add_argument("--bisect", action="store_true", help="per-tensor error of EVERY net's gradient in backward order (logit layer first), "
                    "product fp32 beside the fp32 oracle, both against fp64: where along the backward pass the product's extra error enters")
    ap.add_argument("--resolve", type=int, default=0, help="after the fp32 product arm: re-evaluate the fp32 ORACLE with up to N near-kink "
                    "pre-activations put on the other side of their kink (oracle/conditioning.match_kink_sides) and report how close that "
                    "brings it to the product - whether the product's extra distance from fp64 is a handful of flipped masks or arithmetic")
    a = ap.parse_args()
    evaluate(a, arms=tuple(a.arms.split(",")), frozen=a.frozen)


def evaluate(args, arms=("fp32", "bf16"), deterministic=False, frozen=False):
    """args: .st, .steps, .cascade. Prints the table and returns {arm: (loss_rel, {net: (relative L2, cos, length ratio)})}.
    frozen=True adds a second comparison of the SAME snapshot, batch and noise with the critics' learning rate set to 0 (oracle
    fp64 / fp32 and the product in fp32; keys "oracle32_frozen", "fp32_frozen"): the generator's gradient of a step is taken through
    the critics as that step's Adam update leaves them, and one ReLU / LeakyReLU mask that round-off flips in a critic's update moves
    the WHOLE generator gradient by 1-4e-2 - in the fp32 oracle as in the product (seen in both). With the update taken out, what is
    compared is the arithmetic of the generator's forward / backward and of the critics' scoring pass."""
    st, im = args.st, 5 * args.st
    results = {}
    from cpcsv import runtime
    from oracle.cpcsv_oracle import NoiseTape, make_state, pororo_cfg, synthetic_batch, train_step
    from tests import parity_util as pu
    oc = pororo_cfg(st_batch=st, im_batch=im, cascade=args.cascade)
    torch.set_num_threads(min(32, max(1, os.cpu_count() or 1)))      # the oracle's sweet spot on the 256-core hosts (profiles/r05_cpu_threads.txt)
    names = ("G", "D_im", "D_st", "D_se")
    nets_of = lambda s_: (s_.netG, s_.netD_im, s_.netD_st, s_.netD_se)

    # (1) train in fp32 from the oracle's seeded init (so every arm shares one initial state)
    state32 = make_state(oc, seed=0)
    sds0 = {k: copy.deepcopy(n.state_dict()) for k, n in zip(names, nets_of(state32))}
    # deterministic=True: the training steps in the reproducible-reduction mode (no float atomics), so that the state they reach -
    # and with it the bf16 error measured on it, which varies 0.06-0.19 in relative L2 between ordinary runs - is the same every run
    was_det = runtime.deterministic()
    runtime.set_deterministic(bool(deterministic) or was_det)
    tr = pu.make_trainer(oc, sds0, "fp32")
    torch.manual_seed(123)
    torch.cuda.manual_seed_all(123)
    batches = [tuple(pu.to_dev(b) for b in synthetic_batch(oc, seed=10 + i)) for i in range(8)]
    t0 = time.time()
    hist = []
    for i in range(args.steps):
        out = tr.train_step(*batches[i % len(batches)])
        if i % 50 == 0 or i == args.steps - 1:
            hist.append((i, float(out["G/loss"]), float(out["img_D/loss"]), float(out["st_D/loss"]), float(out["seg_D/loss"])))
    torch.cuda.synchronize()
    print("# %s model, ST=%d IM=%d, cfg/final.yml widths: %d fp32 product steps in %.1f s; (step, G, img_D, st_D, seg_D losses): %s"
          % ("cascade" if args.cascade else "plain", st, im, args.steps, time.time() - t0,
             "; ".join("%d: %.3f %.3f %.3f %.3f" % h for h in hist)))
    # (2) snapshot
    sds = {k: {n_: v.detach().cpu().clone() for n_, v in net.state_dict().items()} for k, net in zip(names, tr.nets)}
    runtime.set_deterministic(was_det)
    del tr
    torch.cuda.empty_cache()
    stb, imb = synthetic_batch(oc, seed=999)

    # (3) fp64 oracle step (and the fp32 oracle on the same noise)
    torch.set_default_dtype(torch.float64)
    try:
        st64 = make_state(oc, seed=0)
        for k, n in zip(names, nets_of(st64)):
            n.load_state_dict(sds[k])
        d = lambda b: {k: v.double() for k, v in b.items()}
        torch.manual_seed(5)
        t0 = time.time()
        ref64 = train_step(st64, d(stb), d(imb), noise=NoiseTape())
        print("# fp64 oracle step: %.1f s" % (time.time() - t0))
    finally:
        torch.set_default_dtype(torch.float32)
    tape = [t.float() for t in ref64["noise_tape"]]
    st32 = make_state(oc, seed=0)
    for k, n in zip(names, nets_of(st32)):
        n.load_state_dict(sds[k])
    ref32 = train_step(st32, stb, imb, noise=NoiseTape(tape))

    def detail(tag, grads, key="G", gk="grads_G", top=12):
        """which tensors carry the whole-vector error of one net: share of the squared error, own relative error"""
        want = ref64[gk]
        den = sum(float((g.double() ** 2).sum()) for g in want.values())
        rows = []
        for n, g in want.items():
            e = float(((grads[key][n].double().cpu() - g.double()) ** 2).sum())
            rows.append((e / den, (e / max(float((g.double() ** 2).sum()), 1e-300)) ** 0.5, float((g.double() ** 2).sum()) / den, n))
        rows.sort(reverse=True)
        tot = sum(r[0] for r in rows)
        print("#   %s, net %s: whole-vector relative L2 %.3g; tensors by share of the squared error (share, own relative L2, share of |g|^2)"
              % (tag, key, tot ** 0.5))
        for sh, own, gsh, n in rows[:top]:
            print("#     %-44s %6.1f %%   %.3g   %6.2f %%" % (n, 100 * sh / max(tot, 1e-300), own, 100 * gsh))

    def bisect(grads):
        """every weight tensor of every net in BACKWARD order (the layer next to the loss first): own relative L2 error against fp64 of
        the fp32 oracle and of the product, their ratio; the first tensor along the backward pass where the product is > 10x the oracle
        is where its extra error enters (errors made there ride on into every layer in front of it)."""
        for key, gk in pu.NETKEYS:
            want = ref64[gk]
            print("#   bisect %s (backward order): tensor, |g64|, oracle fp32 rel, product fp32 rel, ratio" % key)
            first = None
            for n in reversed(list(want)):
                g = want[n].double()
                gn = float((g ** 2).sum()) ** 0.5
                if gn == 0.0 or g.numel() < 64:
                    continue
                eo = float(((ref32[gk][n].double() - g) ** 2).sum()) ** 0.5 / gn
                ep = float(((grads[key][n].double().cpu() - g) ** 2).sum()) ** 0.5 / gn
                mark = ""
                if first is None and ep > 10 * max(eo, 1e-7):
                    first, mark = n, "   <-- first tensor with product > 10x oracle"
                print("#     %-52s %9.3g  %9.2e  %9.2e  %7.1f%s" % (n, gn, eo, ep, ep / max(eo, 1e-12), mark))

    def resolve(grads, most):
        """product fp32 against the fp32 ORACLE (not fp64): before and after the oracle's near-kink elements are put on the product's side"""
        from oracle import conditioning as COND
        stc = make_state(oc, seed=0)
        for k, n in zip(names, nets_of(stc)):
            n.load_state_dict(sds[k])
        snap = pu.oracle_snapshot(stc)

        def dist(out):
            tot = {}
            for key, gk in pu.NETKEYS:
                num = sum(float(((grads[key][n].double().cpu() - g.double()) ** 2).sum()) for n, g in out[gk].items())
                den = sum(float((g.double() ** 2).sum()) for g in out[gk].values())
                tot[key] = (num / max(den, 1e-300)) ** 0.5
            return tot
        before = dist(ref32)
        t0 = time.time()
        out, _, kept = COND.match_kink_sides(oc, snap, stb, imb, tape, lambda o: sum(dist(o).values()), limit=64.0, most=most)
        after = dist(out)
        print("#   resolve: product fp32 vs the fp32 oracle, relative L2 of each net's gradient: as evaluated  %s" % "  ".join("%s %.2e" % kv for kv in before.items()))
        print("#            with %d near-kink element(s) of the oracle put on the other side (%.0f s): %s" % (len(kept), time.time() - t0, "  ".join("%s %.2e" % kv for kv in after.items())))
        for name, numel, safety in kept:
            print("#            flipped: %s (%d elements in the tensor), %.2f round-offs from zero" % (name, numel, safety))

    def against64(grads):
        rows = {}
        for key, gk in pu.NETKEYS:
            want = ref64[gk]
            num = sum(float(((grads[key][n].double().cpu() - g.double()) ** 2).sum()) for n, g in want.items())
            den = sum(float((g.double() ** 2).sum()) for g in want.values())
            dot = sum(float((grads[key][n].double().cpu() * g.double()).sum()) for n, g in want.items())
            n1 = sum(float((grads[key][n].double() ** 2).sum()) for n in want) ** 0.5
            rows[key] = ((num / max(den, 1e-300)) ** 0.5, dot / max(n1 * den ** 0.5, 1e-300), n1 / max(den ** 0.5, 1e-300))
        return rows

    def losses(out, prod):
        lnames = dict(pu.LOSS_NAMES)
        if args.cascade:
            lnames.update(pu.CASCADE_NAMES)
        return max(abs(float(out[pk if prod else rk]) - float(ref64[rk])) / (abs(float(ref64[rk])) + 1e-8) for rk, pk in lnames.items())

    print("%-16s %-9s %9s   %s" % ("arm", "loss_rel", "", "per net: relative L2 / cos / |g|/|g64| of the whole gradient vector vs the fp64 oracle"))
    o32 = against64({key: ref32[gk] for key, gk in pu.NETKEYS})
    results["oracle32"] = (losses(ref32, False), o32)
    print("%-16s %-9.2e %9s   %s" % ("oracle fp32", losses(ref32, False), "", "  ".join("%s %.3g/%.4f/%.3f" % ((k,) + v) for k, v in o32.items())))
    if getattr(args, "detail", False):
        detail("oracle fp32", {key: ref32[gk] for key, gk in pu.NETKEYS})
        # the fp32 oracle with the PRODUCT's BatchNorm statistics: per-tile (128 rows) sums and sums of squares in fp32, combined in
        # double, var = E[x^2] - E[x]^2 (csrc/norm.hip bn_finalize_kernel); everything else stays torch's fp32
        import torch.nn.functional as TF
        real_bn = TF.batch_norm

        def bn_emul(input, running_mean, running_var, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
            if not training or input.dtype != torch.float32:
                return real_bn(input, running_mean, running_var, weight, bias, training, momentum, eps)
            x = input
            c = x.shape[1]
            xr = x.movedim(1, -1).reshape(-1, c)
            n = xr.shape[0]
            pad = (-n) % 128
            xp = TF.pad(xr, (0, 0, 0, pad))
            s_t = xp.view(-1, 128, c).sum(1)
            q_t = (xp * xp).view(-1, 128, c).sum(1)
            s, q = s_t.double().sum(0), q_t.double().sum(0)
            mu = s / n
            var = (q / n - mu * mu).clamp_min(0)
            invstd = (1.0 / torch.sqrt(var + eps)).float()
            muf = mu.float()
            with torch.no_grad():
                if running_mean is not None:
                    running_mean.mul_(1 - momentum).add_(momentum * muf.detach())
                    running_var.mul_(1 - momentum).add_(momentum * (var.detach() * n / max(n - 1, 1)).float())
            scale = weight * invstd
            shift = bias - muf * scale
            shp = [1, c] + [1] * (x.dim() - 2)
            return x * scale.view(shp) + shift.view(shp)
        st32e = make_state(oc, seed=0)
        for k, n in zip(names, nets_of(st32e)):
            n.load_state_dict(sds[k])
        TF.batch_norm = bn_emul
        try:
            ref32e = train_step(st32e, stb, imb, noise=NoiseTape(tape))
        finally:
            TF.batch_norm = real_bn
        o32e = against64({key: ref32e[gk] for key, gk in pu.NETKEYS})
        print("%-16s %-9.2e %9s   %s" % ("oracle32+prodBN", losses(ref32e, False), "", "  ".join("%s %.3g/%.4f/%.3f" % ((k,) + v) for k, v in o32e.items())))
        detail("oracle fp32 with the product's BatchNorm statistics", {key: ref32e[gk] for key, gk in pu.NETKEYS})
    # (4) product arms
    was = runtime.set_deterministic(True)
    try:
        for dtype in arms:
            trp = pu.make_trainer(oc, sds, dtype)
            pu.set_noise(trp.nets[0], pu.TapeSource(tape))
            grads = {}
            hooks = pu._capture_grads(trp, grads)
            out = trp.train_step(pu.to_dev(stb), pu.to_dev(imb))
            torch.cuda.synchronize()
            for h in hooks:
                h()
            rows = against64(grads)
            results[dtype] = (losses(out, True), rows)
            if getattr(args, "detail", False):
                detail("product " + dtype, grads)
            if getattr(args, "bisect", False) and dtype == "fp32":
                bisect(grads)
            if getattr(args, "resolve", 0) and dtype == "fp32":
                resolve(grads, args.resolve)
            print("%-16s %-9.2e %9s   %s" % ("product " + dtype, losses(out, True), "", "  ".join("%s %.3g/%.4f/%.3f" % ((k,) + v) for k, v in rows.items())))
            del trp, grads
            torch.cuda.empty_cache()
    finally:
        runtime.set_deterministic(was)
    if frozen:
        ocf = oc.but(d_lr=0.0)
        torch.set_default_dtype(torch.float64)
        try:
            st64 = make_state(ocf, seed=0)
            for k, n in zip(names, nets_of(st64)):
                n.load_state_dict(sds[k])
            d = lambda b: {k: v.double() for k, v in b.items()}
            ref64 = train_step(st64, d(stb), d(imb), noise=NoiseTape([t.double() for t in tape]))       # (against64 / losses read this name)
        finally:
            torch.set_default_dtype(torch.float32)
        st32 = make_state(ocf, seed=0)
        for k, n in zip(names, nets_of(st32)):
            n.load_state_dict(sds[k])
        ref32 = train_step(st32, stb, imb, noise=NoiseTape(tape))
        o32 = against64({key: ref32[gk] for key, gk in pu.NETKEYS})
        results["oracle32_frozen"] = (losses(ref32, False), o32)
        print("%-16s %-9.2e %9s   %s" % ("oracle32 d_lr=0", losses(ref32, False), "", "  ".join("%s %.3g/%.4f/%.3f" % ((k,) + v) for k, v in o32.items())))
        was = runtime.set_deterministic(True)
        try:
            trp = pu.make_trainer(ocf, sds, "fp32")
            pu.set_noise(trp.nets[0], pu.TapeSource(tape))
            grads = {}
            hooks = pu._capture_grads(trp, grads)
            out = trp.train_step(pu.to_dev(stb), pu.to_dev(imb))
            torch.cuda.synchronize()
            for h in hooks:
                h()
            rows = against64(grads)
            results["fp32_frozen"] = (losses(out, True), rows)
            print("%-16s %-9.2e %9s   %s" % ("product32 d_lr=0", losses(out, True), "", "  ".join("%s %.3g/%.4f/%.3f" % ((k,) + v) for k, v in rows.items())))
            del trp, grads
            torch.cuda.empty_cache()
        finally:
            runtime.set_deterministic(was)
    return results


if __name__ == "__main__":
    main()
