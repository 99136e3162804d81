#!/usr/bin/env python3
"""Per-kernel roofline table from the files tools/collect_profiles.sh writes (gpurun_out/<round>/):
   python tools/roofline_table.py gpurun_out/r02 > profiles/r02_roofline.txt
Peaks (MI355X_MICROARCH.md): 2500 TFLOP/s dense bf16 MFMA, 8 TB/s HBM3E. GEMM rows: algorithmic FLOPs of the reference's
algorithm (9 taps on the upsampled map) / HIP-event time of the launch (bench.py's meter, eager launches after the timed
region). Streaming rows: HBM bytes per launch from the PMC passes (read = 2 x FETCH_SIZE, write = WRITE_SIZE) / average
duration of the same kernel in the PMC pass; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)."""
import ast
import re
import sys

d = sys.argv[1]
steps_metered = 10
print("# MI355X roofline table, bench workload (ST=12 IM=60, cfg/final.yml widths, bf16); peaks: 2500 TFLOP/s MFMA bf16, 8 TB/s HBM")
print()
print("## 1. MFMA gather-GEMM family by shape (bench.py meter; (kind, M, N, taps, Cs, ...)); frac = TF/s / 2500")
print("%-46s %9s %8s %9s %7s" % ("shape", "calls/step", "avg_us", "TFLOP/s", "frac"))
tot_ms = 0.0
rows = []
for line in open(d + "/gemm_by_shape.txt"):
    m = re.match(r"(\(.*?\))\s+n=\s*(\d+)\s+ms=\s*([\d.]+)\s+TF/s=\s*([\d.]+)", line)
    if not m:
        continue
    shape, n, ms, tf = m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))
    rows.append((ms, shape, n, tf))
    tot_ms += ms
for ms, shape, n, tf in sorted(rows, reverse=True)[:48]:
    print("%-46s %9.1f %8.1f %9.1f %7.3f" % (shape, n / steps_metered, 1e3 * ms / n, tf, tf / 2500.0))
print("(%d shapes, %.2f ms of GEMM time per step in the meter's eager launches)" % (len(rows), tot_ms / steps_metered))
print()
mf = {}
try:
    for line in open(d + "/pmc_mfma.txt"):
        if line.startswith("#") or line.startswith("kernel"):
            continue
        p = line.split()
        name = " ".join(p[:-6])
        calls, avg_us, busy = int(p[-6]), float(p[-5]), float(p[-4])
        mf[name] = busy / calls / (1024 * avg_us * 1e-6 * 2.4e9)
except OSError:
    pass
print("## 2. every kernel with > 0.1 ms per step: HBM bytes per launch (PMC) / duration; frac = TB/s / 8")
print("%-60s %7s %8s %8s %8s %7s %6s %9s" % ("kernel", "calls", "avg_us", "read_MB", "write_MB", "TB/s", "frac", "MFMA-busy"))
for line in open(d + "/pmc_traffic.txt"):
    if line.startswith("#") or line.startswith("kernel"):
        continue
    p = line.split()
    if len(p) < 5:
        continue
    name = re.sub(r"^void ", "", " ".join(p[:-4]))
    calls, rd, wr, us = int(p[-4]), float(p[-3]), float(p[-2]), float(p[-1])
    tb = (rd + wr) / us                      # MB / us = TB/s
    print("%-60s %7d %8.1f %8.2f %8.2f %7.2f %6.3f %9s" % (name[:60], calls, us, rd, wr, tb, tb / 8.0,
                                                          ("%.3f" % mf[name]) if name in mf else "-"))
print()
print("## 3. streaming (thin) convolution kernels IN THE STEP: algorithmic bytes per launch (every tensor once; bench.py's meter reads")
print("##    them off the tensors of each launch) / average duration of that kernel in the step's kernel trace (graph replays)")
import json
thin_b = {}
try:
    line = [l for l in open(d + "/bench_default.json") if l.startswith("{")][-1]
    for k, v in json.loads(line)["roofline"]["streaming_convs_hbm"].items():
        thin_b[k] = v["MB_per_launch"] * 1e6
except (OSError, KeyError, IndexError, ValueError):
    pass
kmap = {"thin3x3_fwd_taps_kernel<128, 16, 8>": "thin3x3_fwd_c128", "thin3x3_fwd_roll_kernel<128>": "thin3x3_fwd_c128",
        "thin3x3_fwd_roll_kernel<64>": "thin3x3_fwd_c64", "thin3x3_fwd_taps_kernel<64, 16, 8>": "thin3x3_fwd_c64",
        "thin3x3_dgrad_kernel<128>": "thin3x3_dgrad_c128", "thin3x3_dgrad_kernel<64>": "thin3x3_dgrad_c64",
        "thin3x3_wgrad_rows_kernel<128>": "thin3x3_wgrad_c128", "thin3x3_wgrad_rows_kernel<64>": "thin3x3_wgrad_c64",
        "thin4x4s2_fwd_kernel<128>": "thin4x4s2_fwd_c128", "thin4x4s2_dgrad_kernel": "thin4x4s2_dgrad_c128",
        "thin4x4s2_wgrad_kernel": "thin4x4s2_wgrad_c128"}
print("%-40s %7s %8s %8s %8s %7s" % ("kernel", "calls", "avg_us", "MB", "TB/s", "frac"))
try:
    for line in open(d + "/kernel_stats.txt"):
        if line.startswith("#") or line.startswith("kernel"):
            continue
        p = line.split()
        if len(p) < 5:
            continue
        name = re.sub(r"^void ", "", " ".join(p[:-4]))
        if name in kmap and kmap[name] in thin_b:
            us, b = float(p[-2]), thin_b[kmap[name]]
            print("%-40s %7d %8.1f %8.1f %8.2f %7.3f" % (name, int(p[-4]), us, b / 1e6, b / us / 1e6, b / us / 1e6 / 8.0))
    print("(thin_slab_reduce_kernel, the fixed-order sum of the weight-gradient slabs, is a separate row of section 2)")
except OSError:
    pass
