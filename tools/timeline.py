#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace rocpd sqlite: per step (delimited by the generator's adam_kernel,
the last kernel of a step) the wall time, the UNION of kernel intervals (time the GPU runs at least one kernel), the
sum of kernel durations, the average concurrency, and the idle gaps; plus the same split per kernel family.
   python tools/timeline.py <results.db> [skip_steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start_c = "start" if "start" in cols else "start_timestamp"
end_c = "end" if "end" in cols else "end_timestamp"
rows = list(db.execute("select name, %s, %s, queue_id from kernels order by %s" % (start_c, end_c, start_c))) if "queue_id" in cols \
    else [(n, s, e, 0) for n, s, e in db.execute("select name, %s, %s from kernels order by %s" % (start_c, end_c, start_c))]
clean = lambda n: re.sub(r"^void ", "", re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n)))
# step boundaries: every 4th adam_kernel END (se, im, st, G optimisers; G's is the last launch of a step)
adam_ends = sorted(e for n, s, e, q in rows if "adam_kernel" in n)
bounds = adam_ends[3::4]
print("# %d kernels, %d steps in the trace; first %d steps skipped" % (len(rows), len(bounds), skip))


def family(n):
    n = clean(n)
    if n.startswith(("gemm_nt", "wgrad_tn", "gemm_epilogue", "conv_patch")):
        return "gemm"
    if n.startswith("thin_"):
        return "thin-conv"
    if n.startswith("bn_"):
        return "batchnorm"
    if n.startswith("sn_"):
        return "spectral"
    if "pack" in n or "layer_update" in n:
        return "pack/unpack/update"
    if "adam" in n:
        return "adam"
    if n.startswith("at::") or n.startswith("__amd"):
        return "torch/runtime"
    return "other"


tot_wall = tot_union = tot_sum = 0.0
fam = {}
gaps = []
nsteps = 0
for i in range(max(skip, 1), len(bounds)):
    t0, t1 = bounds[i - 1], bounds[i]
    ks = [(n, s, e) for n, s, e, q in rows if s >= t0 and e <= t1 + 1]
    if not ks:
        continue
    nsteps += 1
    tot_wall += (t1 - t0)
    tot_sum += sum(e - s for n, s, e in ks)
    cur_s, cur_e = None, None
    union = 0
    for n, s, e in sorted(ks, key=lambda k: k[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
                gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    tot_union += union
    for n, s, e in ks:
        f = fam.setdefault(family(n), [0, 0.0])
        f[0] += 1
        f[1] += e - s
ms = lambda x: x / 1e6 / max(nsteps, 1)
print("per step: wall %.3f ms | GPU busy (union of kernels) %.3f ms | sum of kernel durations %.3f ms | concurrency %.2f | idle %.3f ms"
      % (ms(tot_wall), ms(tot_union), ms(tot_sum), tot_sum / max(tot_union, 1), ms(tot_wall - tot_union)))
gaps.sort(reverse=True)
print("idle gaps per step: n=%.0f, >20us: %.0f, >5us: %.0f ; largest (us): %s" % (
    len(gaps) / max(nsteps, 1), sum(g > 20e3 for g in gaps) / max(nsteps, 1), sum(g > 5e3 for g in gaps) / max(nsteps, 1),
    [round(g / 1e3, 1) for g in gaps[:8]]))
print("%-22s %9s %12s" % ("family", "calls/step", "ms/step(sum)"))
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-22s %9.1f %12.3f" % (k, n / max(nsteps, 1), ms(d)))

# ---- optional: what runs in the LAST `win` ms of a step (the generator's backward pass), per queue and per kernel
if len(sys.argv) > 3:
    win = float(sys.argv[3]) * 1e6
    perq, perk = {}, {}
    n = 0
    for i in range(max(skip, 1), len(bounds)):
        t1 = bounds[i]
        t0 = t1 - win
        n += 1
        for nme, s, e, q in rows:
            if s >= t0 and e <= t1 + 1:
                a = perq.setdefault(q, [0, 0.0]); a[0] += 1; a[1] += e - s
                b = perk.setdefault((q, clean(nme)[:60]), [0, 0.0]); b[0] += 1; b[1] += e - s
    print("# last %.1f ms of each step, per queue: launches/step, busy ms/step" % (win / 1e6))
    for q, (c, d) in sorted(perq.items(), key=lambda kv: -kv[1][1]):
        print("queue %s: %.0f launches, %.3f ms" % (q, c / n, d / 1e6 / n))
        for (qq, k), (c2, d2) in sorted(((kk, vv) for kk, vv in perk.items() if kk[0] == q), key=lambda kv: -kv[1][1])[:12]:
            print("      %-60s %6.1f %8.3f" % (k, c2 / n, d2 / 1e6 / n))
