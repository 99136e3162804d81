"""The critics' forward / data-gradient GEMMs ALONE under every tile shape x split-K factor: is the policy (gemm.hip pick_nt +
cpcsv.kernels.plan_splitk) leaving time on the table for the few-tile / long-K shapes (head conv 3x3 on 4x4 maps, tower conv 3 / 4)?
One child process per tile shape (CPCSV_NT_FORCE is read when the library loads); per layer and kind the time of the policy's own
choice and of each forced split factor.   python tools/tile_probe.py   (GPU box)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))

LAYERS = [  # name, cin, cout, map, batch, k, s, p
    ("head 3x3 1481->992 @4x4 n=179", 1481, 992, 4, 179, 3, 1, 1),
    ("head 3x3 1481->992 @4x4 n=60 (scoring)", 1481, 992, 4, 60, 3, 1, 1),
    ("tower4 4x4s2 496->992 @8x8 n=120", 496, 992, 8, 120, 4, 2, 1),
    ("tower3 4x4s2 248->496 @16x16 n=120", 248, 496, 16, 120, 4, 2, 1),
    ("tower2 4x4s2 124->248 @32x32 n=120", 124, 248, 32, 120, 4, 2, 1),
]
TILES = {"policy": None, "128x128": 0, "128x64": 1, "256x128": 4}
SPLITS = (0, 1, 2, 3, 4, 5, 6, 8)     # 0 = the policy's own


def child():
    import torch
    from cpcsv import functional as F, kernels as K, modules as M, runtime, _lib as L
    runtime.set_compute_dtype("bf16")
    dev = "cuda"
    torch.manual_seed(0)
    for name, cin, cout, hw, batch, k, s, p in LAYERS:
        net = M.FusedSequential(M.Conv2d(cin, cout, k, s, p, bias=False, spectral=False), M.BatchNorm2d(cout), torch.nn.LeakyReLU(0.2)).to(dev)
        x = torch.randn(batch, cin, hw, hw, device=dev)
        h = F.ToNhwcFn.apply(x, runtime.tdtype()).detach().requires_grad_()
        y = net(h)
        y.backward(torch.randn_like(y))
        torch.cuda.synchronize()
        for lay in net._plan():
            for key, d in getattr(lay, "descs", {}).items():
                if not isinstance(key, tuple) or key[0] not in ("fwd", "dgrad") or not isinstance(d, L.GemmDesc):
                    continue
                rows = (d.M // (d.MH * d.MW)) * d.OH * d.OW if d.scatter else (d.M // 4 if d.pool_rows else d.M)   # gemm.hip out_rows()
                own = max(1, d.splitk)
                res = []
                for sk in SPLITS:
                    want = own if sk == 0 else sk
                    if want > 1:
                        ldws = (d.N + 7) // 8 * 8
                        ws = torch.empty((want, rows, ldws), dtype=torch.float32, device=dev)
                        d.splitk, d.ws, d.ldws, d.ws_rows = want, ws.data_ptr(), ldws, rows
                    else:
                        d.splitk, d.ws = 1, None
                    try:
                        for _ in range(3):
                            K.gemm_nt(d)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(30):
                            K.gemm_nt(d)
                        e1.record()
                        torch.cuda.synchronize()
                        res.append("%s%d:%.1f" % ("*" if sk == 0 else "", want, e0.elapsed_time(e1) / 30 * 1e3))
                    except Exception as e:
                        res.append("%d:err" % want)
                        torch.cuda.synchronize()
                print("ROW|%s|%s|M=%d N=%d taps=%d Cs=%d|%s" % (name, key[0], d.M, d.N, d.ntaps, d.Cs, " ".join(res)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    print("# us per launch alone (bf16); columns split:us, * = the policy's split for that tile shape")
    for tname, force in TILES.items():
        env = dict(os.environ)
        if force is not None:
            env["CPCSV_NT_FORCE"] = str(force)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        rows = [l for l in out.stdout.splitlines() if l.startswith("ROW|")]
        if not rows:
            print("tile %s failed: %s" % (tname, out.stderr[-400:]))
        for l in rows:
            _, name, kind, shape, res = l.split("|")
            print("%-8s %-42s %-5s %-34s %s" % (tname, name, kind, shape, res))


if __name__ == "__main__":
    main()
