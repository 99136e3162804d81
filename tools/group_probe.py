"""Would ONE grouped launch across the three critics beat the three concurrent launches the step issues today?

The three critics run identical tower shapes on three streams. For every tower layer (4x4 stride-2 conv, bf16) and each of its
three GEMM launches (forward, data gradient, weight gradient) this measures, per "three critics' worth" of work:
  alone x3   : one critic's launch (120 images: real | fake) timed alone, times three (what three SERIAL launches would cost)
  3 streams  : three critics' launches issued on three streams at once (separate weights / activations), wall time per round
  grouped    : ONE launch over 360 images (the hardware throughput a grouped launch can reach: same tile count, same bytes; the
               per-group weight pointer it would need does not change the main loop)
usage: python tools/group_probe.py   (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpcstoryvisualization-pytorch_amd"))
import torch  # noqa: E402
from cpcsv import functional as F, modules as M, runtime  # noqa: E402
from cpcsv import kernels as K, _lib as L  # noqa: E402

runtime.set_compute_dtype("bf16")
dev = "cuda"


def build(cin, cout, hw, batch, k=4, s=2, p=1):
    net = M.FusedSequential(M.Conv2d(cin, cout, k, s, p, bias=False, spectral=False), M.BatchNorm2d(cout), torch.nn.LeakyReLU(0.2)).to(dev)
    x = torch.randn(batch, cin, hw, hw, device=dev)
    h = F.ToNhwcFn.apply(x, runtime.tdtype()).detach().requires_grad_()
    y = net(h)
    y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
    fns = {}
    for lay in net._plan():
        for key, d in getattr(lay, "descs", {}).items():
            if not isinstance(key, tuple) or key[0] not in ("fwd", "dgrad", "wgrad"):
                continue
            if isinstance(d, L.GemmDesc):
                fns[key[0]] = (lambda d=d: K.gemm_nt(d))
            elif isinstance(d, L.WgradDesc):
                fns[key[0]] = (lambda d=d: K._call("cpcsv_wgrad_tn", K.C.byref(d), K.stream()))
    return net, h, y, fns          # (keep everything alive: the descriptors point into these tensors)


def time_one(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def time_streams(fns, streams, reps=30):
    main = torch.cuda.current_stream()
    for s_, fn in zip(streams, fns):
        with torch.cuda.stream(s_):
            fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s_ in streams:
        s_.wait_stream(main)
    for _ in range(reps):
        for s_, fn in zip(streams, fns):
            with torch.cuda.stream(s_):
                fn()
    for s_ in streams:
        main.wait_stream(s_)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    torch.manual_seed(0)
    streams = [torch.cuda.Stream() for _ in range(3)]
    print("layer (cin->cout @map)   launch |  alone x3   3 streams   grouped(360)  us per three critics | TF/s: alone, 3 streams, grouped")
    for hw, cin, cout in ((32, 124, 248), (16, 248, 496), (8, 496, 992)):
        one = [build(cin, cout, hw, 120) for _ in range(3)]
        big = build(cin, cout, hw, 360)
        gf3 = 3 * 2.0 * 120 * (hw // 2) ** 2 * cout * cin * 16 / 1e9
        for kind in ("fwd", "dgrad", "wgrad"):
            if kind not in one[0][3]:
                continue
            alone = 3 * time_one(one[0][3][kind])
            conc = time_streams([o[3][kind] for o in one], streams)
            grp = time_one(big[3][kind])
            tf = lambda u: gf3 / u * 1e3
            print("%4d->%4d @%2dx%-2d        %-6s | %8.1f %10.1f %12.1f                       | %7.0f %7.0f %7.0f"
                  % (cin, cout, hw, hw, kind, alone, conc, grp, tf(alone), tf(conc), tf(grp)), flush=True)
        del one, big
        torch.cuda.empty_cache()
