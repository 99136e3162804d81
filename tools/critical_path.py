#!/usr/bin/env python3
"""Critical-path view of one step of a rocprofv3 --kernel-trace rocpd sqlite.

The step is cut into its phases at recognisable kernels (first thin4x4s2_fwd = the critics start; last bce/mlsm of the
scoring passes; the generator's adam_kernel = end) and, inside every phase, the longest dependency chain is recovered by
walking BACKWARDS from the phase's last-finishing kernel: the predecessor of a kernel is the previous kernel of its own
queue when that one ended at most 25 us before it started (stream order), otherwise the kernel on ANY queue that finished
last before it started (an event join: the producer it waited for) - so the walk follows the chain the phase waited on. Printed per phase: wall time, the chain's kernel time
by family, the summed gaps, and the chain itself (offset, duration, gap before, queue, kernel).
   python tools/critical_path.py <results.db> [step_from_end=1] [min_us_listed=8]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
min_list = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start_c = "start" if "start" in cols else "start_timestamp"
end_c = "end" if "end" in cols else "end_timestamp"
rows = list(db.execute("select name, %s, %s, queue_id from kernels order by %s" % (start_c, end_c, start_c)))
clean = lambda n: re.sub(r"^void ", "", re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n)))
adam_ends = sorted(r[2] for r in rows if "adam_kernel" in r[0])
bounds = adam_ends[3::4]
t0, t1 = bounds[-1 - back], bounds[-back]
ks = [(clean(n), s, e, q) for n, s, e, q in rows if s >= t0 and e <= t1 + 1]
qs = sorted({k[3] for k in ks})


def family(n):
    if n.startswith(("gemm_nt", "wgrad_tn", "gemm_epilogue", "conv_patch")):
        return "gemm"
    if n.startswith("thin"):
        return "thin-conv"
    if n.startswith("bn_"):
        return "batchnorm"
    if n.startswith("sn_"):
        return "spectral"
    if "pack" in n or "layer_update" in n or "adam" in n:
        return "update/pack"
    if n.startswith(("at::", "__amd")):
        return "torch/runtime"
    return "other"


first_critic = min((s for n, s, e, q in ks if n.startswith("thin4x4s2_fwd")), default=t0)
losses = [e for n, s, e, q in ks if n.startswith(("bce", "mlsm"))]
# scoring passes: the LAST burst of loss kernels (after the critic updates)
score_end = max(losses) if losses else t1
phases = [("no-grad generator pass (fakes for the critics)", t0, first_critic),
          ("critic updates || generator forward, then scoring", first_critic, score_end),
          ("generator backward + updates", score_end, t1)]
print("# step of %.3f ms, %d kernels; %d queues" % ((t1 - t0) / 1e6, len(ks), len(qs)))
for title, a, b in phases:
    seg = [k for k in ks if k[2] > a and k[1] < b + 1 and k[2] <= b + 1]
    if not seg:
        continue
    seg.sort(key=lambda k: k[2])
    chain = [seg[-1]]
    while True:
        cur = chain[-1]
        prev = [k for k in seg if k[2] <= cur[1] and k is not cur]
        if not prev:
            break
        same = [k for k in prev if k[3] == cur[3]]
        p = max(same, key=lambda k: k[2]) if same else None
        if p is None or cur[1] - p[2] > 25000:
            p = max(prev, key=lambda k: k[2])
        chain.append(p)
    chain.reverse()
    fam, gaps, ktime = {}, 0.0, 0.0
    for i, (n, s, e, q) in enumerate(chain):
        fam[family(n)] = fam.get(family(n), 0.0) + (e - s) / 1e3
        ktime += (e - s) / 1e3
        if i:
            gaps += max(0.0, (s - chain[i - 1][2]) / 1e3)
    print("\n## %s: wall %.3f ms; chain of %d kernels = %.3f ms of kernel time + %.3f ms of gaps" % (title, (b - a) / 1e6, len(chain), ktime / 1e3, gaps / 1e3))
    print("   chain time by family (us): " + ", ".join("%s %.0f" % kv for kv in sorted(fam.items(), key=lambda kv: -kv[1])))
    allk = sum((e - s) for n, s, e, q in seg) / 1e3
    print("   all kernels of the phase: %d, %.3f ms summed (concurrency %.2f)" % (len(seg), allk / 1e3, allk / max((b - a) / 1e3, 1e-9)))
    small = [c for c in chain if (c[2] - c[1]) / 1e3 < min_list]
    print("   chain kernels shorter than %.0f us: %d (%.3f ms with their gaps)" % (min_list, len(small), sum((c[2] - c[1]) for c in small) / 1e6))
    print("   offset_us  dur_us  gap_us  queue  kernel   (kernels >= %.0f us)" % min_list)
    for i, (n, s, e, q) in enumerate(chain):
        if (e - s) / 1e3 >= min_list:
            gap = (s - chain[i - 1][2]) / 1e3 if i else 0.0
            print("   %9.1f %7.1f %7.1f  %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, qs.index(q), n[:72]))
