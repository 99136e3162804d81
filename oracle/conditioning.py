"""How far a fixture's pre-activations are from their ReLU / LeakyReLU kinks, in units of fp32 round-off. TEST INFRASTRUCTURE ONLY
(used by oracle/gen_golden.py when it picks the fixtures' seeds, and by tests/test_oracle_vs_golden.py, which re-derives the
stored figures from the committed fixtures).

Why: one pre-activation within round-off of zero lands on either side of its kink depending on summation order; its
derivative (1 against 0.2 / 0) then moves the gradient of its layer and of everything in front of it. In a layer of n elements
one flipped element is worth ~1/sqrt(n) of the layer's gradient (a random-signed sum of n terms loses one of them): 1.8e-2 for
the story critic's head at ST=3 (3 x 16 positions x 64 channels - exactly the event round 5's lock-step run met), 2.8e-3 for a
64x64 map of the generator. The single-step gradient band of the parity tests is 5e-3, so a layer is SENSITIVE when
1/sqrt(n) > 5e-3, i.e. n < 40 000 elements.

What is measured: the oracle runs the step from the same state in fp32, in fp64, and three more times in fp64 with every weight and
input perturbed by one fp32 ulp (relative, random), with a tap on the input of every activation (reference model.py:33,45,78,263,
288,500-513,542-555,584-597, cascade_model.py:33,40,315). For every element
    safety = |z64| / max over the four non-exact evaluations of |z - z64|
is the number of fp32 round-off errors between the element and its kink; an element whose fp32 value carries the other sign than
its fp64 value is a FLIP. Distance relative to the tensor's maximum is the wrong
yardstick: the fake images of a tiny-width generator hold tanh(~0) pixels, the critics' first conv turns them into
pre-activations of 1e-9 of the tensor maximum whose own round-off is 1e-16 of it - harmless - while a BatchNorm output of 1e-7
of the maximum is one ulp from its kink. A fixture is WELL-CONDITIONED when no sensitive layer of any net, call and step has a
flip or an element with safety < SAFETY_MIN. (Requiring 1e-4 of the tensor maximum on EVERY element, as first planned, is not
reachable by any seed: a step evaluates 1.3 M pre-activations, a N(0,1)-shaped population puts 3e-4 of them within 1e-4 of zero;
the best of 60 seed pairs had 1.3e-7 on the sensitive layers alone.)
"""
import copy

import torch
import torch.nn as nn

SENSITIVE_NUMEL = 40000
SAFETY_MIN = 16.0


class KinkTap:
    """Pre-hooks on every nn.ReLU / nn.LeakyReLU of the oracle's nets (+ the output of CA_NET.fc, whose ReLU is functional):
    the inputs of all activation calls of a step, in call order, as float64. `force` = {call index: [(flat index, side)]}
    additionally puts those pre-activations on the given side of the kink (+-1e-30: the value is unchanged for every purpose
    but the derivative the activation takes there)."""

    def __init__(self, state, force=None, record=True):
        self.rows, self.handles, self.calls, self.force, self.record = [], [], 0, force or {}, record
        for tag, net in (("G", state.netG), ("D_im", state.netD_im), ("D_st", state.netD_st), ("D_se", state.netD_se)):
            if net is None:
                continue
            for name, mod in net.named_modules():
                if isinstance(mod, (nn.ReLU, nn.LeakyReLU)):
                    self.handles.append(mod.register_forward_pre_hook(self._pre(tag + "." + name)))
                elif name.endswith("ca_net.fc") or name == "ca_net.fc":
                    self.handles.append(mod.register_forward_hook(self._post(tag + "." + name)))

    def _see(self, name, z):
        k, self.calls = self.calls, self.calls + 1
        if self.record:
            self.rows.append((name, z.detach().double().flatten().clone()))
        todo = self.force.get(k)
        if not todo:
            return None
        z = z.clone()
        flat = z.view(-1)
        for idx, side in todo:
            flat[idx] = 1e-30 if side > 0 else -1e-30
        return z

    def _pre(self, name):
        def fn(mod, inp):
            z = self._see(name, inp[0])
            return None if z is None else (z,) + tuple(inp[1:])
        return fn

    def _post(self, name):
        return lambda mod, inp, out: self._see(name, out)

    def close(self):
        for h in self.handles:
            h.remove()


def _to64(state):
    """A float64 copy of a TrainState: weights, buffers, Adam moments and step counts."""
    from .cpcsv_oracle import make_state
    keep = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        with torch.random.fork_rng(devices=[]):          # (make_state seeds the global generator: the caller's noise stream stays untouched)
            st64 = make_state(state.cfg)
        dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
        pairs = ((state.netG, st64.netG, state.optG, st64.optG), (state.netD_im, st64.netD_im, state.optD_im, st64.optD_im),
                 (state.netD_st, st64.netD_st, state.optD_st, st64.optD_st), (state.netD_se, st64.netD_se, state.optD_se, st64.optD_se))
        for net, net64, opt, opt64 in pairs:
            net64.load_state_dict({k: dbl(v) for k, v in net.state_dict().items()})
            osd = copy.deepcopy(opt.state_dict())
            osd["state"] = {i: {k: dbl(v) for k, v in s.items()} for i, s in osd["state"].items()}
            opt64.load_state_dict(osd)
    finally:
        torch.set_default_dtype(keep)
    return st64


PROBES = 3          # perturbed fp64 evaluations per step (see kink_safety)
PROBE_EPS = 1.2e-7  # their relative perturbation: one fp32 ulp


def _perturb(state64, batches, k):
    """Multiply every weight and every floating-point input by (1 + PROBE_EPS N(0,1)), reproducibly (probe k)."""
    g = torch.Generator().manual_seed(7700 + k)
    with torch.no_grad():
        for net in (state64.netG, state64.netD_im, state64.netD_st, state64.netD_se):
            if net is None:
                continue
            for p in net.parameters():
                p.mul_(1.0 + PROBE_EPS * torch.randn(p.shape, generator=g, dtype=torch.float64))
    out = []
    for b in batches:
        out.append({key: (v * (1.0 + PROBE_EPS * torch.randn(v.shape, generator=g, dtype=torch.float64)) if v.is_floating_point() and key != "labels" else v)
                    for key, v in b.items()})
    return out


def kink_safety(state, st_batch, im_batch, tape=None, shuffle=None, near=None, near_limit=0.0, probes=PROBES):
    """Runs ONE oracle step from `state` in fp32 (in place: `state` advances, like any train_step), the same step in fp64 from a copy
    of the state taken before, and `probes` more fp64 evaluations whose weights and inputs are perturbed by one fp32 ulp (relative,
    random): an element's ROUND-OFF SCALE is the largest deviation from the fp64 value that any of these evaluations shows - the
    realised fp32 error alone is one draw of a random quantity (the same step evaluated in another process gave 0.6 where the
    first gave 14), the ulp-sized perturbations sample what any other correct fp32 evaluation may do to that element, including
    elements whose inputs are tiny (their error is tiny with them). tape = None: the fp32 run draws its noise from the global
    generator (seed it first) and the fp64 runs replay those draws. Returns (rows, out32): rows = [(layer, numel, min safety, flips)]
    per activation call, safety = |z64| / round-off scale. `near` (a list) additionally receives every element with
    safety < near_limit as (safety, call index, flat index, the side the fp32 run took, layer, numel)."""
    from .cpcsv_oracle import NoiseTape, train_step
    st64 = _to64(state)
    probe_states = [_to64(state) for _ in range(probes)]
    tap32 = KinkTap(state)
    out32 = train_step(state, st_batch, im_batch, noise=NoiseTape(tape), shuffle=shuffle)
    tape = out32["noise_tape"]
    tap32.close()
    keep = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        dbl = lambda b: {k: (v.double() if v.is_floating_point() else v) for k, v in b.items()}
        tape64 = [t.double() for t in tape]
        tap64 = KinkTap(st64)
        train_step(st64, dbl(st_batch), dbl(im_batch), noise=NoiseTape(tape64), shuffle=shuffle)
        tap64.close()
        scale = [(z32 - z64).abs() for (_, z32), (_, z64) in zip(tap32.rows, tap64.rows)]
        for k, stp in enumerate(probe_states):
            stb, imb = _perturb(stp, (dbl(st_batch), dbl(im_batch)), k)
            tapp = KinkTap(stp)
            train_step(stp, stb, imb, noise=NoiseTape(tape64), shuffle=shuffle)
            tapp.close()
            for i, ((_, zp), (_, z64)) in enumerate(zip(tapp.rows, tap64.rows)):
                scale[i] = torch.maximum(scale[i], (zp - z64).abs())
            del tapp
    finally:
        torch.set_default_dtype(keep)
    assert [r[0] for r in tap32.rows] == [r[0] for r in tap64.rows]
    rows = []
    for call, ((name, z32), (_, z64)) in enumerate(zip(tap32.rows, tap64.rows)):
        err = scale[call]
        live = (err > 0) & ((z64 != 0) | (z32 != 0))
        ratio = torch.where(live, z64.abs() / err.clamp_min(1e-300), torch.full_like(err, float("inf")))
        flips = int(((z32 > 0) != (z64 > 0)).sum())
        rows.append((name, int(z64.numel()), float(ratio.min()), flips))
        if near is not None:
            for idx in torch.nonzero(ratio < near_limit).flatten().tolist():
                near.append((float(ratio[idx]), call, idx, 1 if z32[idx] > 0 else -1, name, int(z64.numel())))
    return rows, out32


def summary(rows):
    """(smallest safety over the sensitive layers, flips in them, the worst sensitive row, smallest safety over all layers, flips in all)."""
    sens = [r for r in rows if r[1] < SENSITIVE_NUMEL]
    worst = min(sens, key=lambda r: r[2])
    return worst[2], sum(r[3] for r in sens), worst, min(r[2] for r in rows), sum(r[3] for r in rows)


# ---- the lock-step tests' answer to a near-kink element ----------------------------------------------------------------------
def state_from_snapshot(cfg, snap):
    """TrainState from [(net state_dict, optimiser state_dict)] x 4 (tests/parity_util.oracle_snapshot)."""
    from .cpcsv_oracle import make_state
    with torch.random.fork_rng(devices=[]):
        st = make_state(cfg)
    for (nsd, osd), net, opt in zip(snap, (st.netG, st.netD_im, st.netD_st, st.netD_se), (st.optG, st.optD_im, st.optD_st, st.optD_se)):
        net.load_state_dict(nsd)
        opt.load_state_dict(copy.deepcopy(osd))
    return st


def match_kink_sides(cfg, snap, st_batch, im_batch, tape, error_of, shuffle=None, limit=SAFETY_MIN, most=12):
    """The oracle's step from `snap`, re-evaluated with near-kink pre-activations put on the OTHER side of their kink, until it
    agrees with what `error_of` measures against (the product's step). The candidates are the elements that THIS host's fp32
    evaluation leaves fewer than `limit` of its own round-off errors away from zero (at most `most`, closest first) - the only
    elements about which two correct fp32 evaluations of the step may disagree. Greedy: a candidate's flip is kept when it
    lowers error_of(out) by more than a fifth. Returns (out, state after the step, [(layer, numel, safety)] of the kept flips);
    the caller applies its ordinary, tight tolerances to `out` - nothing is loosened, the product must equal the reference
    arithmetic for ONE assignment of sides to the listed elements."""
    from .cpcsv_oracle import NoiseTape, train_step
    near = []
    kink_safety(state_from_snapshot(cfg, snap), st_batch, im_batch, tape, shuffle, near=near, near_limit=limit)
    near.sort()
    near = near[:most]

    def run(force):
        st = state_from_snapshot(cfg, snap)
        tap = KinkTap(st, force=force, record=False)
        try:
            out = train_step(st, st_batch, im_batch, noise=NoiseTape(tape), shuffle=shuffle)
        finally:
            tap.close()
        return out, st

    force, kept = {}, []
    out, st = run(force)
    err = error_of(out)
    for safety, call, idx, side, name, numel in near:
        trial = {k: list(v) for k, v in force.items()}
        trial.setdefault(call, []).append((idx, -side))
        out2, st2 = run(trial)
        err2 = error_of(out2)
        if err2 < 0.8 * err:
            force, out, st, err = trial, out2, st2, err2
            kept.append((name, numel, safety))
    return out, st, kept
