"""cpcsv_layer_update (csrc/norm.hip layer_update_kernel: accumulator -> gradient -> Adam -> operand copies, 32 B per parameter)
ALONE on the step's biggest weight shapes: microseconds, TB/s, and the same with one phase of the kernel compiled out of the
launch (CPCSV_UPD_PROBE bits: 1 no accumulator pass, 2 no Adam pass, 4 no forward copy, 8 no data-gradient copy) - where the
time of the optimiser stream goes.   python tools/update_probe.py   (on the GPU box; spawns one child per probe value)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))

SHAPES = [  # name, Cout, Cin, taps, S, sum, dense
    ("G.upsample1 (sub-pixel 2048->1024)", 1024, 2048, 9, 16, 1, 0),
    ("G.upsample2 (sub-pixel 1024->512)", 512, 1024, 9, 16, 1, 0),
    ("G.seg_c (3x3 1024->2048)", 2048, 1024, 9, 9, 0, 0),
    ("G.fc (dense 613->32768)", 32768, 613, 1, 1, 0, 1),
    ("D.head conv3x3 (1481->992)", 992, 1481, 9, 9, 0, 0),
    ("D.enc3 (4x4 496->992)", 992, 496, 16, 16, 0, 0),
]


def child():
    import torch
    from cpcsv import _lib as L, kernels as K
    from cpcsv.functional import SUB_MASKS
    from cpcsv.runtime import pad8
    dev = "cuda"
    rows = []
    for name, cout, cin, taps, S, sm, dense in SHAPES:
        cin_s, cout_s = pad8(cin), pad8(cout)
        G = torch.randn(cout, S * cin_s, device=dev) * 1e-3
        p = torch.randn(cout, cin, taps, device=dev) * 0.02
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        fwd = torch.empty(cout, S * cin_s, dtype=torch.bfloat16, device=dev)
        bwd = None if dense else torch.empty(cin_s, S * cout_s, dtype=torch.bfloat16, device=dev)
        lin = torch.empty(S * cin_s, cout_s, dtype=torch.bfloat16, device=dev) if dense else None
        hyper = torch.tensor([3.0, 1e-4], device=dev)
        d = L.UpdateDesc()
        d.G, d.p, d.m, d.v = G.data_ptr(), p.data_ptr(), m.data_ptr(), v.data_ptr()
        d.fwd, d.bwd, d.lin = fwd.data_ptr(), (bwd.data_ptr() if bwd is not None else None), (lin.data_ptr() if lin is not None else None)
        d.hyper = hyper.data_ptr()
        d.beta1, d.beta2, d.eps = 0.5, 0.999, 1e-8
        d.dtype, d.Cout, d.Cin, d.taps, d.S, d.Cin_s, d.Cout_s, d.sum = L.BF16, cout, cin, taps, S, cin_s, cout_s, sm
        for i in range(L.MAX_TAPS):
            d.tapmap[i] = i if i < S else -1
            d.masks[i] = SUB_MASKS[i] if (sm and i < 16) else 0
        d.gscale, d.step_add = 1.0, 0.0
        for _ in range(3):
            K.layer_update(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.layer_update(d)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        n = cout * cin * taps
        nbytes = 4 * cout * S * cin_s + 24 * n + 2 * cout * S * cin_s + 2 * S * cin_s * cout_s
        rows.append((name, us, nbytes / us / 1e6))
    print(";".join("%s|%.1f|%.2f" % r for r in rows))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    print("# cpcsv_layer_update alone, bf16 operand copies; us (TB/s of its algorithmic bytes); probe: 1 no accumulator pass, 2 no Adam pass, "
          "4 no forward copy, 8 no data-gradient copy")
    table = {}
    probes = (0, 1, 2, 4, 8, 12, 14, 13)
    for pr in probes:
        env = dict(os.environ, CPCSV_UPD_PROBE=str(pr))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if "|" in l]
        if not line:
            print("probe %d failed: %s" % (pr, out.stderr[-500:]))
            continue
        for item in line[-1].split(";"):
            name, us, tb = item.split("|")
            table.setdefault(name, {})[pr] = (float(us), float(tb))
    print("%-40s " % "layer" + " ".join("%12s" % ("probe %d" % p) for p in probes))
    for name, row in table.items():
        print("%-40s " % name + " ".join(("%7.1f/%4.2f" % row[p]) if p in row else "%12s" % "-" for p in probes))


if __name__ == "__main__":
    main()
