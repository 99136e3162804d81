"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the
product nets carry the reference's state_dict keys, host logic (config, geometry, init) behaves,
and the product path refuses to run without the GPU instead of falling back."""
import os
import re
import types

import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests import parity_util as pu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from cpcsv import _lib
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "cpcsv_hip.h")).read()
    declared = set(re.findall(r"\b(cpcsv_[a-z0-9_]+)\s*\(", header))
    declared -= {"cpcsv_tap", "cpcsv_gemm_desc", "cpcsv_wgrad_desc", "cpcsv_sn_job", "cpcsv_bn_groups", "cpcsv_update_desc", "cpcsv_scalar_list", "cpcsv_copy_list", "cpcsv_logit_groups", "cpcsv_wgrad_piece", "cpcsv_wgrad_target", "cpcsv_small_wgrad_list", "cpcsv_cond_head", "cpcsv_cond_head_grad"}
    assert declared, "no symbols parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.cpcsv_arch() == b"gfx950"
    # constants the Python side mirrors
    consts = dict(re.findall(r"#define\s+(CPCSV_[A-Z_]+)\s+(\d+)", header))
    assert int(consts["CPCSV_BN_SUM_COPIES"]) == _lib.BN_SUM_COPIES
    assert int(consts["CPCSV_MAX_TAPS"]) == _lib.MAX_TAPS


def _header_struct_fields(header, name):
    """Field names of `typedef struct <name> { ... } <name>;` in declaration order (comments stripped; `a, b;` lists split)."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.search(r"(\w+)\s*(\[[^\]]*\])*$", first.strip()).group(1))
        names += [re.search(r"(\w+)", r).group(1) for r in rest]
    return names


def test_abi_layout_matches_ctypes():
    """sizeof / offsetof of every descriptor struct as the LIBRARY was compiled (cpcsv_abi_layout) == the ctypes mirrors
    in cpcsv/_lib.py == the field list of include/cpcsv_hip.h. A field added on one side only fails here (and at load())."""
    import ctypes as C
    from cpcsv import _lib
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "cpcsv_hip.h")).read()
    cnames = {0: "cpcsv_tap", 1: "cpcsv_gemm_desc", 2: "cpcsv_wgrad_desc", 3: "cpcsv_sn_job", 4: "cpcsv_bn_groups", 5: "cpcsv_update_desc", 6: "cpcsv_scalar_list", 7: "cpcsv_copy_list", 8: "cpcsv_logit_groups", 9: "cpcsv_wgrad_piece", 10: "cpcsv_wgrad_target", 11: "cpcsv_small_wgrad_list",
              12: "cpcsv_pack_job", 13: "cpcsv_pack_list", 14: "cpcsv_txt_job", 15: "cpcsv_txt_stage", 16: "cpcsv_cond_head", 17: "cpcsv_cond_head_grad"}
    buf = (C.c_int * 256)()
    for which, struct in _lib.ABI_STRUCTS.items():
        need = lib.cpcsv_abi_layout(which, None, 0)
        assert need == 2 + 2 * len(struct._fields_), (struct.__name__, need)
        n = lib.cpcsv_abi_layout(which, buf, len(buf))
        assert list(buf[:n]) == _lib.layout_of(struct), struct.__name__
        assert [f[0] for f in struct._fields_] == _header_struct_fields(header, cnames[which]), struct.__name__
    assert lib.cpcsv_abi_layout(99, buf, len(buf)) < 0
    assert lib.cpcsv_abi_layout(1, buf, 3) < 0                      # buffer too small
    # a mirror with a field missing is caught
    class Short(C.Structure):
        _fields_ = _lib.GemmDesc._fields_[:-1]
    saved = _lib.ABI_STRUCTS[1]
    _lib.ABI_STRUCTS[1] = Short
    try:
        with pytest.raises(RuntimeError, match="ABI mismatch"):
            _lib.verify_layout(lib)
    finally:
        _lib.ABI_STRUCTS[1] = saved


def test_integration_md_struct_stubs_are_current():
    """The ctypes stub in INTEGRATION.md is generated from cpcsv/_lib.py (tools/gen_integration_stub.py): regenerate on drift."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_integration_stub.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize("tag", ["plain", "cascade", "seq"])
def test_state_dict_keys_equal_reference(tag):
    fx = gu.load("step_%s.npz" % tag)
    oc = gu.cfg_of(fx)
    nets = pu.product_nets(oc)
    sds = gu.state_dicts(fx)
    for net, key in zip(nets, ("G", "D_im", "D_st", "D_se")):
        ref = sds[key]
        res = net.load_state_dict(ref, strict=True)          # identical key set and shapes
        assert not res.missing_keys and not res.unexpected_keys
        sd = net.state_dict()
        assert list(sd.keys()) == list(ref.keys()) or set(sd.keys()) == set(ref.keys())
        for k in ref:
            assert torch.equal(sd[k].cpu(), ref[k]), k


def test_weights_init_dispatch_matches_reference_rules():
    """Conv*/BatchNorm*/Linear by class NAME (miscc/utils.py:191-201); GRUCell untouched; spectral convs hit weight_orig."""
    from miscc.utils import weights_init
    oc = gu.cfg_of(gu.load("step_plain.npz"))
    torch.manual_seed(0)
    G, D_im, _, _ = pu.product_nets(oc)
    assert abs(G.upsample1[1].weight.std().item() - 0.02) < 0.004
    assert abs(G.upsample1[2].weight.mean().item() - 1.0) < 0.02 and G.upsample1[2].bias.abs().max() == 0
    assert abs(G.fc[0].weight.std().item() - 0.02) < 0.002
    sn = D_im.encode_img[2]
    assert "weight_orig" in dict(sn.named_parameters()) and abs(sn.weight_orig.std().item() - 0.02) < 0.004
    assert G.recurrent.weight_ih.abs().max().item() <= 1.0 / np.sqrt(G.motion_dim) + 1e-6
    assert abs(G.recurrent.weight_ih.std().item() - 0.02) > 0.005      # NOT N(0,.02)


def test_product_refuses_cpu_execution():
    oc = gu.cfg_of(gu.load("step_plain.npz"))
    G = pu.product_nets(oc)[0]
    with pytest.raises(RuntimeError, match="GPU|HIP|cuda"):
        G.sample_images(torch.zeros(4, oc.motion_dim), torch.zeros(4, oc.video_len, oc.text_dim))


def test_conv_geometry_plans():
    from cpcsv.functional import ConvGeom
    g = ConvGeom(3, 1, 1, up=1)
    assert g.out_hw(4, 4) == (8, 8) and len(g.fwd_taps()) == 9
    (taps, mh, mw, pool, scatter), = g.dgrad_launches(4, 4)
    assert (mh, mw, pool, scatter) == (8, 8, 1, None) and (1, 1, 0) in taps and (-1, -1, 8) in taps
    g = ConvGeom(4, 2, 1)
    assert g.out_hw(64, 64) == (32, 32)
    plans = g.dgrad_launches(64, 64)
    assert len(plans) == 4 and all(len(p[0]) == 4 for p in plans) and g.dgrad_covers_all()
    covered = set()
    for taps, mh, mw, pool, (ih, iw, sy, sx, py, px) in plans:
        covered.add((py, px))
        for oy, ox, wt in taps:                                   # every tap obeys  y + pad - u = 2*oy'
            u, v = divmod(wt, 4)
            assert (py + 1 - u) % 2 == 0 and (py + 1 - u) // 2 == oy and (px + 1 - v) // 2 == ox
    assert covered == {(0, 0), (0, 1), (1, 0), (1, 1)}
    g = ConvGeom(3, 2, 1)                                          # cascade downBlock
    assert sorted(len(p[0]) for p in g.dgrad_launches(8, 8)) == [1, 2, 2, 4]


def test_config_merge_rules(tmp_path):
    from miscc import config as C
    snap = dict(C.cfg.TRAIN)
    y = tmp_path / "c.yml"
    y.write_text("TRAIN:\n  ST_BATCH_SIZE: 7\nVIDEO_LEN: 4\n")
    C.cfg_from_file(str(y))
    assert C.cfg.TRAIN.ST_BATCH_SIZE == 7 and C.cfg.VIDEO_LEN == 4
    y.write_text("NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        C.cfg_from_file(str(y))
    y.write_text("VIDEO_LEN: 'five'\n")
    with pytest.raises(ValueError):
        C.cfg_from_file(str(y))
    C.cfg.TRAIN.ST_BATCH_SIZE, C.cfg.VIDEO_LEN = snap["ST_BATCH_SIZE"], 5


def test_get_multi_acc_numpy_api():
    from miscc.utils import get_multi_acc
    fx = gu.load("ops.npz")
    assert get_multi_acc(fx["acc/logits"], fx["acc/labels"]) == pytest.approx(float(fx["acc/out"]), rel=1e-12)


def test_sample_dump_sheet_layout(tmp_path):
    """F3 (reference miscc/utils.py:229-281): one row per story, its T frames side by side, generated | ground truth, with
    torchvision.make_grid's 2-pixel padding; values clamp to [-1,1] and map to uint8."""
    from miscc import utils as U
    from miscc.config import cfg
    keep = cfg.VIDEO_LEN
    cfg.VIDEO_LEN = 3
    try:
        vids = torch.full((2, 3, 3, 4, 4), -1.0)          # (B, C, T, H, W)
        vids[0, :, 1] = 1.0                                # story 0, frame 1: white
        vids[1, 0, 2] = 3.0                                # story 1, frame 2: red channel over range -> clamps to 255
        sheet = U.save_story_results(vids, vids, [["a", "b"]] * 3, "000", str(tmp_path))
        row_h, row_w = 4 + 2 * 2, 3 * 4 + 4 * 2            # one story row: frame height + top/bottom pad; 3 frames + 4 pads
        assert sheet.dtype == np.uint8 and sheet.shape == (2 * (row_h + 2) + 2, 2 * (row_w + 4), 3)
        y0, x0 = 2 + 2, 2 + 2                              # outer grid pad + inner grid pad
        assert (sheet[y0:y0 + 4, x0 + 6:x0 + 10] == 255).all() and (sheet[y0:y0 + 4, x0:x0 + 4] == 0).all()
        y1 = y0 + row_h + 2
        assert (sheet[y1:y1 + 4, x0 + 12:x0 + 16, 0] == 255).all() and (sheet[y1:y1 + 4, x0 + 12:x0 + 16, 1] == 0).all()
        assert (sheet[:, : sheet.shape[1] // 2] == sheet[:, sheet.shape[1] // 2:]).all()      # generated | ground truth halves
        assert "a\n" in open(str(tmp_path / "fake_samples_000.txt")).read()
        n = U.save_all_img(vids.clamp(0, 1), 0, str(tmp_path))
        assert n == 6 and (tmp_path / "6.png").exists()
    finally:
        cfg.VIDEO_LEN = keep


def test_near_kink_element_is_resolved_not_tolerated():
    """oracle/conditioning.match_kink_sides (what the fp32 lock-step tests do when a step misses its band, tests/parity_util.py
    resolve_kinks): a stand-in "product" that put the near-kink pre-activation closest to zero on the OTHER side than this host's oracle is
    matched by flipping exactly that element - after which the two agree to round-off - while a product that is wrong in any other way
    (here: one gradient tensor scaled by 1.01) finds no assignment of sides that explains it, and the caller's tight band then fails."""
    from oracle import conditioning as COND
    from oracle.cpcsv_oracle import NoiseTape, train_step
    from tests import parity_util as pu
    fx = gu.load("step_plain.npz")
    oc, st, _ = pu.oracle_state_for(fx)
    stb, imb = gu.batches(fx)
    tape = gu.noise_tape(fx)
    snap = pu.oracle_snapshot(st)
    with pu.oracle_threads(fx):
        near = []
        COND.kink_safety(COND.state_from_snapshot(oc, snap), stb, imb, tape, near=near, near_limit=1e4)
        # the closest-to-zero element of a SMALL critic layer (a flip there is worth ~1 % of the net's gradient)
        pick = min(r for r in near if r[5] < 20000 and r[4].startswith("D_"))
        safety, call, idx, side, name, numel = pick
        nets = ("grads_G", "grads_D_im", "grads_D_st", "grads_D_se")     # (a flip in a critic's scoring pass shows in G's gradient only)
        base = train_step(COND.state_from_snapshot(oc, snap), stb, imb, noise=NoiseTape(tape))
        st2 = COND.state_from_snapshot(oc, snap)
        tap = COND.KinkTap(st2, force={call: [(idx, -side)]}, record=False)
        prod = train_step(st2, stb, imb, noise=NoiseTape(tape))
        tap.close()

        def err_to(target):
            def err(ref):
                tot = 0.0
                for net in nets:
                    num = sum(float(((ref[net][k] - g) ** 2).sum()) for k, g in target[net].items())
                    den = sum(float((g ** 2).sum()) for g in target[net].values())
                    tot += (num / den) ** 0.5
                return tot
            return err
        apart = err_to(prod)(base)
        assert apart > 1e-4, (name, numel, apart)                      # the flip is visible in the gradients
        out, _, kept = COND.match_kink_sides(oc, snap, stb, imb, tape, err_to(prod), limit=safety * 1.5 + 1.0, most=6)
        assert [(k[0], k[1]) for k in kept] == [(name, numel)], kept
        assert err_to(prod)(out) < 1e-5
        wrong = {net: {k: (g * 1.01 if i == 0 else g) for i, (k, g) in enumerate(prod[net].items())} for net in nets}
        out2, _, kept2 = COND.match_kink_sides(oc, snap, stb, imb, tape, err_to(wrong), limit=safety * 1.5 + 1.0, most=6)
        assert err_to(wrong)(out2) > 1e-4                              # not explained by any side assignment: the band would fail


def test_fixtures_are_well_conditioned():
    """Every fixture stores, per step, how many fp32 round-off errors its closest pre-activation of a flip-sensitive layer lies from
    its ReLU / LeakyReLU kink (oracle/conditioning.py; the seeds were searched for it, oracle/gen_golden.py search()). Re-derived
    here from the committed data with the oracle: no mask disagreement between fp32 and fp64 in a sensitive layer, and the stored
    safety is reproduced (the fp32 round-off itself depends on the host's BLAS: within a factor of 4)."""
    from oracle import conditioning as COND
    from tests import parity_util as pu
    floor = {"step_plain.npz": 3.0, "step_clevr.npz": 3.0, "step_cascade.npz": 2.0, "step_seq.npz": 0.5, "steps3_plain.npz": 1.0, "steps3_cascade.npz": 0.3}
    for name, least in floor.items():
        fx = gu.load(name)
        stored = [float(v) for v in np.atleast_1d(fx["meta/kink_safety"])]
        assert all(v >= least for v in stored) and int(np.sum(fx["meta/kink_flips"])) == 0, (name, stored)
        if name.startswith("steps3") or name == "step_seq.npz":
            continue                                                   # (re-derivation of the single-step fixtures keeps the CPU suite short)
        oc, st, _ = pu.oracle_state_for(fx)
        stb, imb = gu.batches(fx)
        with pu.oracle_threads(fx):
            rows, _ = COND.kink_safety(st, stb, imb, gu.noise_tape(fx), shuffle=gu.shuffle_plan_of(fx))
        got, flips = COND.summary(rows)[:2]
        assert flips == 0 and got >= stored[0] / 4.0, (name, got, stored)


def test_oracle_runs_with_the_fixture_thread_count():
    from tests import golden_util as gu
    from tests import parity_util as pu
    fx = gu.load("steps3_plain.npz")
    keep = torch.get_num_threads()
    with pu.oracle_threads(fx):
        assert torch.get_num_threads() == int(fx["meta/seeds"][3])
    assert torch.get_num_threads() == keep


def test_text_stage_descriptor_packing():
    """cpcsv.textpath._Stages: jobs land in the kernarg struct of their stage in call order, at most CPCSV_TXT_MAX_JOBS per stage,
    unused pointer slots NULL (what cpcsv_text_stage validates; include/cpcsv_hip.h cpcsv_txt_job)."""
    from cpcsv import _lib as L
    from cpcsv import textpath as TP
    S = TP._Stages()
    S.add(2, L.TXT_DENSE, (12, 60), N=365, Kd=368, ldx=368, ldw=368, ldy=368, act=1, eps=1e-5, mom=0.1, x=[0x1000, 0x2000], w=0x3000,
          bias=0x4000, y=[0x5000, 0x6000], P=((0x7000,), (0x8000, 0x9000)), Q=(0xA000, 0xB000))
    S.add(2, L.TXT_GRU_FWD, (12,), (5,), Kd=128, ldx=128, ldw=128, ldy=128, A=(124, 376, 3), x=[0x1100], w=0x1200, bias=0x1300, y=[0x1400])
    S.add(0, L.TXT_PREP, (12, 60), (5, 1))
    assert sorted(S.st) == [0, 2] and S.st[2].njobs == 2 and S.st[0].njobs == 1
    j = S.st[2].job[0]
    assert (j.type, j.npass, j.M[0], j.M[1], j.N, j.K, j.ldx, j.ldw, j.ldy, j.act) == (L.TXT_DENSE, 2, 12, 60, 365, 368, 368, 368, 368, 1)
    assert (j.x[0], j.x[1], j.w, j.bias, j.y[0], j.y[1]) == (0x1000, 0x2000, 0x3000, 0x4000, 0x5000, 0x6000)
    assert j.P[0][0] == 0x7000 and j.P[0][1] is None and j.P[1][1] == 0x9000 and j.Q[1] == 0xB000 and j.Q[2] is None
    assert abs(j.eps - 1e-5) < 1e-12 and abs(j.momentum - 0.1) < 1e-7
    g = S.st[2].job[1]
    assert (g.type, g.npass, g.M[0], g.T[0], g.A[0], g.A[1], g.A[2], g.A[7]) == (L.TXT_GRU_FWD, 1, 12, 5, 124, 376, 3, 0)
    assert g.x[1] is None and g.y[1] is None
    for _ in range(L.TXT_MAX_JOBS - 1):
        S.add(0, L.TXT_PREP, (1,), (1,))
    with pytest.raises(RuntimeError):
        S.add(0, L.TXT_PREP, (1,), (1,))


def test_every_device_collective_goes_through_the_one_communication_stream(monkeypatch):
    """cpcsv.dist: ONE communicator, ONE stream, host order (the only configuration NCCL guarantees without further assumptions), and
    no end event of a collective ever on a stream that graphs are captured on - so there is no timing guard left in the module. The
    routing is tested with stand-in streams (no GPU here): a device tensor's collective is issued inside the communication stream's
    context, ordered behind the caller's stream, and the caller's stream continues behind it; host tensors (gloo) call straight
    through."""
    import inspect
    from cpcsv import dist as cd
    src = inspect.getsource(cd)
    assert "sleep(" not in src and "_STEADY" not in src
    log = []

    class FakeStream:
        def __init__(self, name):
            self.name, self.cuda_stream = name, id(self)

        def wait_stream(self, other):
            log.append("%s waits %s" % (self.name, other.name))

    cur, comm = FakeStream("caller"), FakeStream("comm")
    active = [cur]

    class Ctx:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            active.append(self.s)

        def __exit__(self, *a):
            active.pop()
    monkeypatch.setattr(cd.torch.cuda, "current_stream", lambda: active[-1])
    monkeypatch.setattr(cd.torch.cuda, "stream", lambda s: Ctx(s))
    monkeypatch.setattr(cd, "comm_stream", lambda: comm)
    monkeypatch.setattr(cd.dist, "get_backend", lambda group=None: "nccl")
    dev = type("T", (), {"is_cuda": True})()
    cd._sync_collective(lambda a: log.append("collective async=%s on %s" % (a, active[-1].name)), dev)
    assert log == ["comm waits caller", "collective async=False on comm", "caller waits comm"]
    del log[:]
    host = type("T", (), {"is_cuda": False})()
    cd._sync_collective(lambda a: log.append("collective async=%s on %s" % (a, active[-1].name)), host)
    assert log == ["collective async=False on caller"]
