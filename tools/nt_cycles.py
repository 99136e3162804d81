"""Where a block of the gather-GEMM spends its time: builds csrc/ with -DCPCSV_PROBE=8 into /tmp (in-kernel s_memtime stamps of
wave 0 of every block around [issue the next K tile's loads | ds_read + MFMA of the current one | s_waitcnt + barrier]), runs one
tower layer per shape alone (tools/width_probe.py's harness) and prints per-K-tile averages in shader-clock cycles (s_memtime
counts them: a block's cycles against its share of the launch time give the clock the GEMM actually runs at, ~1.5 GHz).
usage: python tools/nt_cycles.py   (on the GPU box; ~1 min for the build)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PKG = os.path.join(ROOT, "cpcstoryvisualization-pytorch_amd")
sys.path.insert(0, PKG)
LIB = "/tmp/libcpcsv_probe.so"
srcs = [os.path.join(PKG, "csrc", f) for f in ("gemm.hip", "norm.hip", "elementwise.hip", "small.hip", "thin.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-shared",
                       "-DCPCSV_PROBE=8", "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", LIB], cwd=os.path.join(PKG, "csrc"))
from cpcsv import _lib as L
L.LIB_PATH = LIB
import torch
from cpcsv import functional as F, kernels as K, modules as M, runtime

runtime.set_compute_dtype("bf16")
lib = L.load()
lib.cpcsv_probe_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()


def probe(cin, cout, hw, batch, k=4, s=2, p=1, reps=5):
    net = M.FusedSequential(M.Conv2d(cin, cout, k, s, p, bias=False, spectral=False), M.BatchNorm2d(cout), torch.nn.LeakyReLU(0.2)).to("cuda")
    x = torch.randn(batch, cin, hw, hw, device="cuda")
    h = F.ToNhwcFn.apply(x, runtime.tdtype()).detach().requires_grad_()
    y = net(h)
    y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
    out = {}
    for lay in net._plan():
        for key, d in getattr(lay, "descs", {}).items():
            if isinstance(key, tuple) and key[0] in ("fwd", "dgrad") and isinstance(d, L.GemmDesc):
                torch.cuda.synchronize()
                lib.cpcsv_probe_read(buf, 1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    K.gemm_nt(d)
                e1.record()
                torch.cuda.synchronize()
                lib.cpcsv_probe_read(buf, 0)
                issue, mma, wait, total, blocks, kt = (int(v) for v in buf[:6])
                mt, nt_ = lib.cpcsv_gemm_mtile(C.byref(d)), lib.cpcsv_gemm_ntile(C.byref(d))
                out[key[0]] = (e0.elapsed_time(e1) / reps * 1e3, blocks // reps, kt / max(blocks, 1), issue / max(kt, 1),
                               mma / max(kt, 1), wait / max(kt, 1), total / max(blocks, 1), "%dx%d" % (mt, nt_))
    return out


print("# per K tile a wave issues 32 MFMAs of 16 cycles (512); the 256x128 tile runs 8 waves (two per SIMD, one block per CU), the 128x128 /")
print("# 128x64 tiles 4 waves (two / three blocks per CU): MFMA-bound = 1024 cycles per K tile and SIMD in all three cases")
print("cin cout   map batch pass | launch us blocks Ktiles/blk | cycles per K tile: issue  mma  wait | cycles per block   tile  (GHz)")
for hw, batch, pairs in ((8, 120, ((512, 1024),)), (16, 120, ((256, 512),)), (32, 120, ((128, 256),)), (64, 30, ((256, 512), (512, 1024))),
                         (128, 30, ((128, 256), (256, 128)))):
    for cin, cout in pairs:
        for name, (us, blocks, ktb, i_, m_, w_, tot, tile) in probe(cin, cout, hw, batch).items():
            bpc = 1 if tile.startswith("256") else (2 if tile == "128x128" else 3)
            rounds = blocks / (256.0 * bpc)
            ghz = "%.2f" % (tot / (us / rounds * 1e3)) if rounds >= 3 else "  - "
            print(f"{cin:4d} {cout:4d} {hw:3d}x{hw:<3d} {batch:3d} {name:5s} | {us:8.1f} {blocks:6d} {ktb:8.1f} | {i_:22.0f} {m_:5.0f} {w_:5.0f} | {tot:12.0f} {tile:>8s}  {ghz}", flush=True)
