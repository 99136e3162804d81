"""Why are the critics' DF_DIM-124 layers (widths 124 * 2^k) 3-4x slower per FLOP than the generator's power-of-two
layers?  Times one tower layer (4x4 stride-2 conv + BatchNorm + LeakyReLU, bf16, forward and forward+backward) at a
width pair and at its power-of-two neighbour, and mixed pairs that change only Cin or only Cout.
usage: python tools/width_probe.py  (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpcstoryvisualization-pytorch_amd"))
import torch
from cpcsv import functional as F, modules as M, runtime

runtime.set_compute_dtype("bf16")
dev = "cuda"


def layer_ms(cin, cout, hw, batch, k=4, s=2, p=1, reps=40):
    """GPU time of the layer's three GEMM launches (each re-launched `reps` times from its cached descriptor)."""
    from cpcsv import kernels as K, _lib as L
    net = M.FusedSequential(M.Conv2d(cin, cout, k, s, p, bias=False, spectral=False), M.BatchNorm2d(cout), torch.nn.LeakyReLU(0.2)).to(dev)
    x = torch.randn(batch, cin, hw, hw, device=dev)
    h = F.ToNhwcFn.apply(x, runtime.tdtype()).detach().requires_grad_()
    y = net(h)
    dy = torch.randn_like(y)
    y.backward(dy)
    torch.cuda.synchronize()
    out = {"sum_y": float(y.float().double().abs().sum()), "sum_dx": float(h.grad.float().double().abs().sum())}
    for lay in net._plan():
        for key, d in getattr(lay, "descs", {}).items():
            if not isinstance(key, tuple) or key[0] not in ("fwd", "dgrad", "wgrad"):
                continue
            if isinstance(d, L.GemmDesc):
                fn = lambda d=d: K.gemm_nt(d)
            elif isinstance(d, L.WgradDesc):
                fn = lambda d=d: K._call("cpcsv_wgrad_tn", K.C.byref(d), K.stream())
            else:
                continue
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[key[0]] = e0.elapsed_time(e1) / reps * 1e3
    m = batch * (hw // s) ** 2
    gf = 2.0 * m * cout * cin * k * k / 1e9
    return out, gf


if __name__ == "__main__":
    torch.manual_seed(0)
    print("cin cout  map batch |  fwd us dgrad us wgrad us | TF/s fwd dgrad wgrad | sum|y| sum|dx| (same seed: equal across kernel choices)")
    for hw, batch, pairs in ((8, 120, ((496, 992), (512, 1024), (496, 1024), (512, 992), (504, 1008), (480, 960))),
                             (16, 120, ((248, 496), (256, 512), (248, 512), (256, 496))),
                             (32, 120, ((124, 248), (128, 256), (128, 248), (124, 256))),
                             # shapes that take the 256x128 tile (>= 256 such tiles): M = 30720 / 122880 rows
                             (64, 30, ((256, 512), (512, 1024))), (128, 30, ((128, 256), (256, 128)))):
        for cin, cout in pairs:
            t, gf = layer_ms(cin, cout, hw, batch)
            f, dg, wg = t.get("fwd", 0), t.get("dgrad", 0), t.get("wgrad", 0)
            tf = lambda u: gf / u * 1e3 if u else 0.0
            print(f"{cin:4d} {cout:4d} {hw:3d}x{hw:<3d} {batch:3d} | {f:7.1f} {dg:8.1f} {wg:8.1f} | {tf(f):8.1f} {tf(dg):6.1f} {tf(wg):6.1f} | "
                  f"{t['sum_y']:.6e} {t['sum_dx']:.6e}", flush=True)
