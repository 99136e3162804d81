"""Captured HIP graphs must train exactly like the eager step (same kernels, same order, device-side Adam counters,
graph-safe RNG). All comparisons run in the library's deterministic-reduction mode (cpcsv_set_deterministic: no
cross-block float atomics), so eager and replayed runs are comparable to round-off instead of to the ~15 % by step 5
that atomics + GAN dynamics used to allow."""
import os

import pytest
import torch

from tests import golden_util as gu
from tests import parity_util as pu

pytestmark = pytest.mark.gpu

PIECES = ("CPCSV_NOGRAD_GRAPH", "CPCSV_CRITIC_GRAPH", "CPCSV_G_GRAPH", "CPCSV_SCORE_GRAPH")
REL, ABS = 2e-2, 1e-4            # the bound the review asked for; measured agreement is recorded in profiles/


@pytest.fixture(autouse=True)
def _deterministic():
    from cpcsv import runtime
    was = runtime.set_deterministic(True)
    yield
    runtime.set_deterministic(was)
    for k in PIECES + ("CPCSV_GRAPH",):
        os.environ.pop(k, None)


def _trainer():
    fx = gu.load("step_plain.npz")
    oc = gu.cfg_of(fx)
    sds = {k: gu.group(fx, "before/" + k) for k in ("G", "D_im", "D_st", "D_se")}
    tr = pu.make_trainer(oc, sds, "fp32")
    return tr, pu.to_dev(gu.batches(fx)[0]), pu.to_dev(gu.batches(fx)[1])


def _snapshot(tr, out):
    h = {k: float(v) for k, v in out.items() if "Acc" not in k}
    h.update({"|grad %s|" % k: b.norm() for k, b in tr._buckets.items()})
    return h


def _weights(tr):
    return [torch.cat([p.detach().flatten() for p in n.parameters()]).cpu() for n in tr.nets]


def _compare(he, hg):
    for i, (a, b) in enumerate(zip(he, hg)):
        for k in a:
            assert b[k] == b[k] and abs(b[k]) != float("inf"), (i, k)
            assert b[k] == pytest.approx(a[k], rel=REL, abs=ABS), (i, k, a[k], b[k])


def _compare_weights(we, wg, lr_steps):
    """Adam moves an entry by <= ~lr per step; entries whose gradient is pure round-off may differ in sign per run."""
    for a, b in zip(we, wg):
        assert (a - b).abs().max().item() <= 2.2 * lr_steps


def _run_whole(graph, steps=6):
    os.environ["CPCSV_GRAPH"] = "1" if graph else "0"
    for k in PIECES:
        os.environ[k] = "0"
    tr, stb, imb = _trainer()
    # fixed noise (same tensors every step, both modes): the comparison must not depend on how the RNG offsets of a
    # captured graph line up with eager draws
    bank = {}

    def fixed_noise(shape):
        if shape not in bank:
            g = torch.Generator().manual_seed(1000 + len(bank))
            bank[shape] = torch.randn(shape, generator=g).cuda()
        return bank[shape]
    pu.set_noise(tr.nets[0], fixed_noise)
    hist = [_snapshot(tr, tr.train_step_graphed(stb, imb)) for _ in range(steps)]
    torch.cuda.synchronize()
    used_graph = tr.__dict__.get("_gs", {}).get("graph") is not None
    bn = [int(m.num_batches_tracked) for m in tr.nets[0].modules() if hasattr(m, "note_batch") and (m._flush() or True)]
    return hist, _weights(tr), used_graph, bn


def test_graph_replay_matches_eager():
    """Whole-step graph (CPCSV_GRAPH=1): 6 steps eager vs 3 eager + capture + replays."""
    he, we, ge, bne = _run_whole(False)
    hg, wg, gg, bng = _run_whole(True)
    assert not ge and gg, "graph path was not exercised"
    _compare(he, hg)
    _compare_weights(we, wg, 6 * 4e-4)
    assert bne == bng                                  # BatchNorm call counters advance under replay too


def _run_pieces(on, steps=8, seed=321):
    """Plain train_step, live RNG (torch's generator: a captured graph advances it exactly like the eager calls do).
    on=True is the DEFAULT launch mode of bench.py / GANTrainer.train(): no-grad pass, critic real/fake+backward,
    generator forward/backward and scoring graphs all captured after 3 eager calls."""
    os.environ["CPCSV_GRAPH"] = "0"
    for k in PIECES:
        os.environ[k] = "1" if on else "0"
    tr, stb, imb = _trainer()
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    hist = []
    for _ in range(steps):
        hist.append(_snapshot(tr, tr.train_step(stb, imb)))
    torch.cuda.synchronize()
    captured = {"nograd": getattr(tr.__dict__.get("_ng"), "captured", False),
                "gen": getattr(tr.__dict__.get("_gg"), "captured", False),
                "critic": all(g.captured for g in tr.__dict__.get("_cg", {}).values()) and bool(tr.__dict__.get("_cg")),
                "score": all(g.captured for g in tr.__dict__.get("_sg", {}).values()) and bool(tr.__dict__.get("_sg"))}
    return hist, _weights(tr), captured


def test_default_piecewise_graphs_match_eager():
    """ALL four capture switches on vs all off, same seeds, live RNG, fp32, 8 steps (3 eager + 5 replayed): every
    loss, the L2 norm of every network's flat gradient buffer and the weights agree step by step. A captured backward
    that read stale packs, lost a weight-gradient branch or wrote the flat gradient buffer wrongly fails here."""
    he, we, ce = _run_pieces(False)
    hg, wg, cg = _run_pieces(True)
    assert not any(ce.values()), ce
    assert all(cg.values()), "default-mode graphs were not all captured: %s" % cg
    _compare(he, hg)
    _compare_weights(we, wg, 8 * 4e-4)


@pytest.mark.parametrize("st,im", [(3, 70), (14, 16)])
def test_mixed_row_counts_small_dense_weight_gradients(st, im):
    """The small fp32 dense layers (text / motion encoders, GRU cells) pick their weight-gradient route per CALL: at most 64 rows
    -> one inline cpcsv_dense_rows_wgrad launch (non-atomic read-modify-write of the master-layout .grad), more rows -> the
    weight-gradient branch of the captured backward. With ST*T <= 64 < IM (3 x 5 = 15 story rows, 70 images) or ST <= 64 < ST*T
    (14 stories, 70 story frames) the two routes hit the SAME gradient in one backward pass; the inline launch must wait for
    the branch's (cpcsv.runtime.note_side_write / wait_side_writes). Captured pieces vs eager, deterministic reductions, same
    seeds: losses, gradient norms and weights agree (a lost contribution is a gross difference, not round-off)."""
    import copy
    from oracle.cpcsv_oracle import make_state, synthetic_batch, tiny_cfg
    oc = tiny_cfg(st_batch=st, im_batch=im)
    state = make_state(oc, seed=0)
    sds = {k: copy.deepcopy(n.state_dict()) for k, n in zip(("G", "D_im", "D_st", "D_se"), (state.netG, state.netD_im, state.netD_st, state.netD_se))}
    stb, imb = synthetic_batch(oc, seed=1)

    def run(on, steps=6):
        os.environ["CPCSV_GRAPH"] = "0"
        for k in PIECES:
            os.environ[k] = "1" if on else "0"
        tr = pu.make_trainer(oc, sds, "fp32")
        a, b = pu.to_dev(stb), pu.to_dev(imb)
        torch.manual_seed(77)
        torch.cuda.manual_seed_all(77)
        hist = [_snapshot(tr, tr.train_step(a, b)) for _ in range(steps)]
        torch.cuda.synchronize()
        return hist, _weights(tr), getattr(tr.__dict__.get("_gg"), "captured", False)

    he, we, ce = run(False)
    hg, wg, cg = run(True)
    assert not ce and cg, "the generator's captured backward was not exercised"
    _compare(he, hg)
    _compare_weights(we, wg, 6 * 4e-4)


def _snapshot_state(tr):
    snap = {"nets": [{k: v.detach().clone() for k, v in n.state_dict().items()} for n in tr.nets], "opts": [], "rng": torch.cuda.get_rng_state()}
    for opt in tr._opt_of.values():
        if opt is None:
            continue
        snap["opts"].append(([(p, {k: v.clone() for k, v in opt.state[p].items() if torch.is_tensor(v)}) for g in opt.param_groups for p in g["params"] if opt.state[p]],
                             [g.get("step", 0) for g in opt.param_groups], {gi: h[0].clone() for gi, h in opt._hypers.items()}))
    return snap


def _restore_state(tr, snap):
    for n, sd in zip(tr.nets, snap["nets"]):
        n.load_state_dict(sd)                                         # in place: captured graphs keep their pointers
    for opt, (states, steps, hypers) in zip([o for o in tr._opt_of.values() if o is not None], snap["opts"]):
        for p, st in states:
            for k, v in st.items():
                opt.state[p][k].copy_(v)
        for g, s_ in zip(opt.param_groups, steps):
            g["step"] = s_
        for gi, h in hypers.items():
            opt._hypers[gi][0].copy_(h)
    torch.cuda.set_rng_state(snap["rng"])
    torch.cuda.synchronize()


def test_benchmarked_mode_gradients_match_eager_deterministic():
    """The launch mode bench.py times - all four capture switches on, atomics on (cpcsv_set_deterministic(0)), live RNG -
    against the eager deterministic mode the oracle parity tests run in, FROM THE SAME STATE: after 3 eager + 1 capturing
    step the trainer's whole state (weights, buffers, Adam moments and counters, RNG) is saved, step 5 is replayed from
    the captured graphs, the state is restored and step 5 runs again eagerly with deterministic reductions. Gradient
    buckets of all four nets agree within 1e-2 relative L2, losses within 1e-3. (Two free-running runs cannot be held to
    that: Adam turns round-off on ~zero gradients into +-lr moves and the trajectories separate) This
    ties the benchmarked configuration to the oracle-checked one directly instead of transitively."""
    from cpcsv import runtime

    def mode(graphs_on, deterministic):
        os.environ["CPCSV_GRAPH"] = "0"
        for k in PIECES:
            os.environ[k] = "1" if graphs_on else "0"
        runtime.set_deterministic(deterministic)

    def step_with_grads(tr, stb, imb):
        grads, restore = {}, []
        for key, opt in tr._opt_of.items():
            if opt is None:
                continue
            orig = opt.step

            def wrapped(closure=None, _k=key, _o=orig):
                b = tr._buckets[_k]
                grads[_k] = torch.cat([t.detach().flatten().double() for t in [b.flat] + list(b.extra)]).cpu()
                return _o()
            opt.step = wrapped
            restore.append((opt, orig))
        out = tr.train_step(stb, imb)
        torch.cuda.synchronize()
        for opt, orig in restore:
            opt.step = orig
        return {k: float(v) for k, v in out.items() if "Acc" not in k}, grads

    mode(True, False)
    tr, stb, imb = _trainer()
    torch.manual_seed(321)
    torch.cuda.manual_seed_all(321)
    for _ in range(4):
        tr.train_step(stb, imb)
    torch.cuda.synchronize()
    cap = all(getattr(tr.__dict__.get(n), "captured", False) for n in ("_ng", "_gg")) and \
        all(g.captured for g in tr.__dict__.get("_cg", {}).values()) and all(g.captured for g in tr.__dict__.get("_sg", {}).values())
    assert cap, "the default-mode graphs were not all captured"
    snap = _snapshot_state(tr)
    lg, gg = step_with_grads(tr, stb, imb)                      # replayed graphs, float atomics
    _restore_state(tr, snap)
    mode(False, True)
    le, ge = step_with_grads(tr, stb, imb)                      # eager, deterministic reductions
    assert set(ge) == set(gg) == {"G", "im", "st", "se"}
    for k in ge:
        rel = float((ge[k] - gg[k]).norm() / ge[k].norm())
        assert rel < 1e-2, (k, rel)
    for k in le:
        assert lg[k] == pytest.approx(le[k], rel=1e-3, abs=1e-5), (k, le[k], lg[k])


def test_order_critic_update_is_captured_with_fresh_shuffles():
    """USE_SEQ_CONSISTENCY: the story critic's update contains create_random_shuffle, whose decisions are made on
    the HOST every step (numpy / python RNGs, reference miscc/utils.py:17-44). They reach the captured pass through
    persistent device index tensors refreshed before every replay (miscc.utils.ShufflePlanBuffers), so the pass is captured
    like the other critics'. Same seeds for torch, numpy and `random`: 7 steps with the pieces captured agree with 7 eager
    steps (every loss incl. the order terms, gradient norms, weights), and the plan really changes from step to step."""
    import random
    import numpy as np
    import miscc.utils as MU

    def run(on, steps=7):
        os.environ["CPCSV_GRAPH"] = "0"
        for k in PIECES:
            os.environ[k] = "1" if on else "0"
        fx = gu.load("step_seq.npz")
        oc = gu.cfg_of(fx)
        tr = pu.make_trainer(oc, gu.state_dicts(fx), "fp32")
        stb, imb = (pu.to_dev(b) for b in gu.batches(fx))
        torch.manual_seed(11)
        torch.cuda.manual_seed_all(11)
        np.random.seed(11)
        random.seed(11)
        plans, hist = [], []
        orig = MU.shuffle_plan

        def spy(b, t, rate=0.5):
            r = orig(b, t, rate)
            plans.append(tuple(map(tuple, r[2])))
            return r
        MU.shuffle_plan = spy
        try:
            for _ in range(steps):
                hist.append(_snapshot(tr, tr.train_step(stb, imb)))
            torch.cuda.synchronize()
        finally:
            MU.shuffle_plan = orig
        cap = {k: g.captured for k, g in tr.__dict__.get("_cg", {}).items()}
        cap.update({"score_" + k: g.captured for k, g in tr.__dict__.get("_sg", {}).items()})
        return hist, _weights(tr), cap, plans

    he, we, ce, pe = run(False)
    hg, wg, cg, pg = run(True)
    assert not any(ce.values()) and cg.get("st") and cg.get("score_st"), (ce, cg)
    assert pe == pg and len(set(pe)) > 1, "shuffle plans: same sequence in both runs, and not constant"
    assert "st_D/order" in he[0] and "G/consistency" in he[0]
    _compare(he, hg)
    _compare_weights(we, wg, 7 * 4e-4)


def test_nograd_pass_graph_matches_eager():
    """Only the no-grad generator pass captured."""
    def run(on):
        os.environ["CPCSV_GRAPH"] = "0"
        for k in PIECES:
            os.environ[k] = "0"
        os.environ["CPCSV_NOGRAD_GRAPH"] = "1" if on else "0"
        tr, stb, imb = _trainer()
        torch.manual_seed(321)
        torch.cuda.manual_seed_all(321)
        hist = [_snapshot(tr, tr.train_step(stb, imb)) for _ in range(6)]
        torch.cuda.synchronize()
        return hist, _weights(tr), getattr(tr.__dict__.get("_ng"), "captured", False)
    he, we, ue = run(False)
    hg, wg, ug = run(True)
    assert not ue and ug, "the captured no-grad pass was not exercised"
    _compare(he, hg)
    _compare_weights(we, wg, 6 * 4e-4)


def test_side_branches_do_not_change_results(monkeypatch):
    """The round-3 scheduling features only move launches between streams: the four text-encoder chains of the no-grad pass
    on four streams (model._TEXT_STREAMS) and the decoder's fused optimiser launches parked on a third branch of the
    generator's backward graph (CPCSV_LATE_UPDATES). With both off the default mode must produce the same losses, gradient
    norms and weights (deterministic reductions, same seeds, 6 steps: 3 eager + 3 replayed) - a lost dependency (a chain
    reading operand copies that are still being rebuilt, an update overtaking the data-gradient GEMM that reads its operand,
    BatchNorm running statistics applied in the wrong order) shows up as a difference here."""
    import model as model_mod
    import trainer as trainer_mod
    monkeypatch.setattr(trainer_mod, "_FUSED_MIN_NUMEL", 256)     # the fixture's tiny layers onto the fused per-layer update path
    runs, late_used = {}, {}
    for tag, text, late in (("default", True, "1"), ("one_stream", False, "0")):
        monkeypatch.setattr(model_mod, "_TEXT_STREAMS", text)
        monkeypatch.setenv("CPCSV_LATE_UPDATES", late)
        made = []
        orig = trainer_mod.GANTrainer._late_stream

        def spy(self, netG, orig=orig, made=made):
            s_ = orig(self, netG)
            made.append(s_ is not None)
            return s_
        monkeypatch.setattr(trainer_mod.GANTrainer, "_late_stream", spy)
        runs[tag] = _run_pieces(True, steps=6, seed=99)
        monkeypatch.setattr(trainer_mod.GANTrainer, "_late_stream", orig)
        late_used[tag] = any(made)
    assert late_used == {"default": True, "one_stream": False}, late_used
    (hd, wd, cd), (ho, wo, co) = runs["default"], runs["one_stream"]
    assert all(cd.values()) and all(co.values()), (cd, co)
    for i, (a, b) in enumerate(zip(ho, hd)):
        for k in a:
            assert b[k] == pytest.approx(a[k], rel=1e-5, abs=1e-6), (i, k, a[k], b[k])
    for a, b in zip(wo, wd):
        assert (a - b).abs().max().item() <= 1e-6
