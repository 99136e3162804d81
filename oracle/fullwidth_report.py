#!/usr/bin/env python3
"""Full-width oracle-vs-reference error report (SURVEY §8(c) "Full-size parity"). TEST INFRASTRUCTURE ONLY.

Runs ONE training step of the imported reference (/root/reference, same shims as gen_golden.py) and of the oracle
restatement at cfg/final.yml widths (ngf 2048, seg 1024, ndf 124, text 356, T=5) from the same seeded weights, batch
and noise, and prints max-abs / relative errors of every loss scalar and every network's gradient. Too large to
commit as a fixture (G alone is 348 MB); the text report is committed as profiles/r02_oracle_vs_reference_fullwidth.txt.

    python oracle/fullwidth_report.py [ST IM] > profiles/r02_oracle_vs_reference_fullwidth.txt
"""
import importlib.util
import io
import contextlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
args = [a for a in sys.argv[1:]]
sys.argv = sys.argv[:1]
spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "gen_golden.py"))
gg = importlib.util.module_from_spec(spec)
with contextlib.redirect_stdout(io.StringIO()):
    spec.loader.exec_module(gg)
import torch  # noqa: E402
from oracle.cpcsv_oracle import NoiseTape, make_state, pororo_cfg, train_step  # noqa: E402

st_b, im_b = (int(args[0]), int(args[1])) if len(args) >= 2 else (12, 60)
torch.set_num_threads(os.cpu_count() or 1)
oc = pororo_cfg(st_batch=st_b, im_batch=im_b)
with contextlib.redirect_stdout(io.StringIO()):
    run = gg.ReferenceRun(oc, 0)
st = make_state(oc)
pairs = (("G", st.netG, run.netG), ("D_im", st.netD_im, run.netD_im), ("D_st", st.netD_st, run.netD_st), ("D_se", st.netD_se, run.netD_se))
for _, net, ref in pairs:
    net.load_state_dict(ref.state_dict(), strict=True)
fx = {}
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    sc = run.step(fx, "", 1, 1234, full=True)
t_ref = time.time() - t0
grp = lambda pre: {k[len(pre):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(pre)}
tape = [torch.from_numpy(fx[k]) for k in sorted(fx) if k.startswith("noise/")]
t0 = time.time()
out = train_step(st, grp("batch/st/"), grp("batch/im/"), noise=NoiseTape(tape))
t_orc = time.time() - t0
print("# oracle (oracle/cpcsv_oracle) vs the imported reference (/root/reference), ONE step at cfg/final.yml widths,")
print("# ST=%d IM=%d, fp32, %d CPU threads; reference step %.1f s, oracle step %.1f s" % (st_b, im_b, torch.get_num_threads(), t_ref, t_orc))
print("%-16s %16s %16s %10s" % ("scalar", "reference", "oracle", "rel.err"))
for k, v in sc.items():
    o = float(out[k])
    print("%-16s %16.8f %16.8f %10.2e" % (k, v, o, abs(o - v) / (abs(v) + 1e-12)))
print("%-8s %14s %14s %14s   %s" % ("net", "grad rel L2", "max abs err", "max |ref|", "worst tensor (rel L2)"))
for key, gk in (("G", "grads_G"), ("D_im", "grads_D_im"), ("D_st", "grads_D_st"), ("D_se", "grads_D_se")):
    num = den = 0.0
    mabs = mref = 0.0
    worst = (0.0, "")
    for n, g in out[gk].items():
        r = torch.from_numpy(fx["grad/%s/%s" % (key, n)]).double()
        d = g.double() - r
        num += float((d * d).sum()); den += float((r * r).sum())
        mabs = max(mabs, float(d.abs().max())); mref = max(mref, float(r.abs().max()))
        rl = float(d.norm() / max(float(r.norm()), 1e-30))
        if rl > worst[0] and float(r.norm()) > 1e-3:
            worst = (rl, n)
    print("%-8s %14.3e %14.3e %14.3e   %s (%.2e)" % (key, (num / den) ** 0.5, mabs, mref, worst[1], worst[0]))
print("# post-step state (parameters after Adam, BN running statistics, SN u/v): max abs difference per net")
for key, net, ref in pairs:
    m = max(float((a.double() - b.double()).abs().max()) for (_, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()))
    print("%-8s %14.3e" % (key, m))
