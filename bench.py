#!/usr/bin/env python3
"""bench.py — story-frames/s of the CP-CSV training step on MI355X (BASELINE.json metric).

    python bench.py                                  (= --gpus 1 --steps 50 --warmup 10)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the reference loop body (trainer.py:252-416): no-grad G forward, three
critic updates, G update, four Adam steps — on synthetic Pororo-shaped batches already resident in
HBM (ST=12 stories x 5 frames + 60 single frames per rank, 64x64, cfg/final.yml widths, bf16
operands / fp32 accumulate, random-init weights). N>1 = one process per GPU, each rank its own
shard, gradients mean-all-reduced over RCCL before every optimiser step (weak scaling).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     — the dominant kernel family (MFMA gather-GEMM): algorithmic FLOP of its launches in the
                 timed region / their summed HIP-event durations, against the 2.5 PFLOP/s dense bf16 peak;
  cpu_baseline — the oracle (CPU fp32 restatement of the reference step) timed on this host's cores on the
                 same workload: 1 warm-up + 2 timed steps (rank 0, N=1 only).
"""
import os as _os

# ROCm 7.2 hipGraph "packet capture" corrupts earlier graphs once a process holds ~2900 kernel nodes (see
# cpcsv/graphs.many_graphs_safe); the switch is read when the HIP runtime initialises, i.e. before torch touches the GPU
if "torch" not in __import__("sys").modules:       # provably before the HIP runtime reads its flags: cpcsv.runtime trusts this marker
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    _os.environ["CPCSV_PACKET_CAPTURE_EARLY"] = _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # kernel arguments in device memory (the ROCm 7.2 default; 0 costs 1.2 ms/step)

import argparse
import gc
import json
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "cpcstoryvisualization-pytorch_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3


DIMS = {"pororo": (5, 356, 9), "clevr": (4, 72, 15)}      # (VIDEO_LEN, TEXT.DIMENSION, LABEL_NUM); CLEVR: datasets/clevr.py:24,38-41,104


def pororo_cfg(st, im, cascade=False, seq=False, dims="pororo"):
    from miscc.config import cfg
    cfg.VIDEO_LEN, cfg.TEXT.DIMENSION, cfg.LABEL_NUM = DIMS[dims]
    cfg.GAN.CONDITION_DIM, cfg.GAN.Z_DIM, cfg.GAN.DF_DIM, cfg.GAN.GF_DIM, cfg.GAN.GF_SEG_DIM = 124, 100, 124, 256, 1024
    cfg.SEGMENT_LEARNING, cfg.SEGMENT_RATIO, cfg.IMAGE_RATIO = True, 1.0, 5.0
    cfg.CASCADE_MODEL, cfg.USE_SEQ_CONSISTENCY, cfg.EVALUATE_FID_SCORE = cascade, seq, False
    cfg.CONSISTENCY_RATIO = 1.0
    cfg.TRAIN.COEFF.KL = 1.0
    cfg.TRAIN.ST_BATCH_SIZE, cfg.TRAIN.IM_BATCH_SIZE = st, im
    cfg.TRAIN.GENERATOR_LR, cfg.TRAIN.DISCRIMINATOR_LR = 1e-4, 4e-4
    cfg.GPU_ID = '0'
    return cfg


def synthetic_batches(st, im, seed, device, dims="pororo"):
    """Batch dicts with the keys trainer.py:254-274 reads."""
    g = torch.Generator().manual_seed(seed)
    t, d, nl = DIMS[dims]

    def labels(*shape):
        lab = (torch.rand(*shape, nl, generator=g) < 0.3).float()
        lab[..., 0] = torch.where(lab.sum(-1) == 0, torch.ones_like(lab[..., 0]), lab[..., 0])
        return lab
    story = {"images": torch.rand(st, 3, t, 64, 64, generator=g) * 2 - 1,
             "description": torch.randn(st, t, d, generator=g), "labels": labels(st, t)}
    image = {"images": torch.rand(im, 3, 64, 64, generator=g) * 2 - 1,
             "images_seg": torch.rand(im, 1, 64, 64, generator=g) * 2 - 1,
             "description": torch.randn(im, d, generator=g),
             "content": torch.randn(im, t, d + nl, generator=g), "labels": labels(im)}
    return ({k: v.to(device) for k, v in story.items()}, {k: v.to(device) for k, v in image.items()})


class GemmMeter:
    """HIP events (torch.cuda.Event on the launch stream) around every gather-GEMM launch."""

    def __init__(self):
        self.records = []     # (flops, start_event, end_event, kind)
        self.thin = []        # (bytes, start_event, end_event, name)
        self.on = False
        self.overhead_ms = 0.0

    def calibrate(self, pairs=200):
        """What a start/end event pair measures with NOTHING between them (the record commands themselves take time
        on the queue): subtracted from every launch so that the durations agree with rocprofv3's kernel trace."""
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(pairs)]
        for s, e in ev:
            s.record()
            e.record()
        torch.cuda.synchronize()
        t = sorted(s.elapsed_time(e) for s, e in ev)
        self.overhead_ms = t[len(t) // 2]
        return self.overhead_ms

    def _ms(self, s, e):
        return max(s.elapsed_time(e) - self.overhead_ms, 1e-4)

    def install(self):
        from cpcsv import kernels as K
        meter = self
        orig_nt, orig_wg = K.gemm_nt, K.wgrad_run

        def nt(desc):
            if not meter.on:
                return orig_nt(desc)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig_nt(desc)
            e.record()
            ex = 2.0 * desc.M * desc.N * desc.ntaps * desc.Cs
            meter.records.append((ex * getattr(desc, "_algo", 1.0), s, e,
                                  ("nt", desc.M, desc.N, desc.ntaps, desc.Cs, desc.up_shift, desc.pool_rows, desc.scatter), ex))

        def wg(d, dY, X, dW, **kw):
            if not meter.on:
                return orig_wg(d, dY, X, dW, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig_wg(d, dY, X, dW, **kw)
            e.record()
            ex = 2.0 * d.M * d.N * d.ntaps * d.Cs
            meter.records.append((ex * getattr(d, "_algo", 1.0), s, e,
                                  ("wgrad", d.M, d.N, d.ntaps, d.Cs, d.up_shift, d.splits, 0), ex))
        K.gemm_nt, K.wgrad_run = nt, wg
        import cpcsv.functional as F
        F.K.gemm_nt, F.K.wgrad_run = nt, wg

        # the streaming (HBM-bound) convolutions of csrc/thin.hip: algorithmic BYTES per launch (input + output once)
        def thin(name, nbytes, width):
            orig = getattr(K, name)

            def hooked(*a):
                if not meter.on:
                    return orig(*a)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                orig(*a)
                e.record()
                meter.thin.append((nbytes(*a), s, e, "%s_c%d" % (name, width(*a))))      # one row per (entry point, wide-tensor width)
            setattr(K, name, hooked)
        nb = lambda *ts: float(sum(t.numel() * t.element_size() for t in ts))
        thin("thin3x3_fwd", lambda x, w, y, *r: nb(x, y), lambda x, *r: x.shape[-1])
        thin("thin3x3_dgrad", lambda dz, w, dx, *r: nb(dz, dx), lambda dz, w, dx, *r: dx.shape[-1])
        thin("thin3x3_wgrad", lambda dz, x, G, slabs, *r: nb(dz, x), lambda dz, x, *r: x.shape[-1])
        thin("thin4x4s2_fwd", lambda x, w, y, *r: nb(x, y), lambda x, w, y, *r: y.shape[-1])
        thin("thin4x4s2_dgrad", lambda dz, w, dx, *r: nb(dz, dx), lambda dz, *r: dz.shape[-1])
        thin("thin4x4s2_wgrad", lambda dz, x, G, slabs, *r: nb(dz, x), lambda dz, *r: dz.shape[-1])

    def thin_summary(self):
        agg = {}
        for b, s, e, name in self.thin:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += self._ms(s, e)
            a[2] += b
        return {k: {"launches": v[0], "avg_us": round(1e3 * v[1] / v[0], 1), "MB_per_launch": round(v[2] / v[0] / 1e6, 2),
                    "TB_per_s": round(v[2] / (v[1] * 1e-3) / 1e12, 3),
                    "frac_of_8TBps": round(v[2] / (v[1] * 1e-3) / 8e12, 3)} for k, v in agg.items() if v[1] > 0}

    def by_shape(self):
        agg = {}
        for f, s, e, key, _ in self.records:
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += self._ms(s, e)
            a[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return ["%-48s n=%4d ms=%8.3f TF/s=%7.1f" % (str(k), v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0)
                for k, v in rows]

    def summary(self):
        tot_f = tot_ex = tot_ms = 0.0
        n = 0
        for f, s, e, _, ex in self.records:
            tot_f += f
            tot_ex += ex
            tot_ms += self._ms(s, e)
            n += 1
        return tot_f, tot_ex, tot_ms, n


def pmc_traffic():
    """(HBM bytes per launch of the dominant kernel family, where the number comes from). Hardware counters cannot be read from
    inside the process: the figure is the rocprofv3 --pmc measurement (FETCH_SIZE and WRITE_SIZE in separate passes, read side
    doubled as MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_summary.py) of THIS command, either taken in the same
    tools/collect_profiles.sh run (CPCSV_PMC_TRAFFIC_JSON, stamped with that run's commit) or the committed profile."""
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("CPCSV_PMC_TRAFFIC_JSON")
    source = "rocprofv3 --pmc passes of the same tools/collect_profiles.sh run"
    if not path:
        import glob
        found = sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_traffic.json")))       # the latest round's committed passes
        path = found[-1] if found else os.path.join(here, "profiles", "r04_pmc_traffic.json")
        source = "profiles/%s (rocprofv3 --pmc passes of this command, committed)" % os.path.basename(path)
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        d = json.load(fh)
    if d.get("head"):
        source += " @ " + str(d["head"])
    return d.get("gemm_family_bytes_per_launch"), source


_GEMM_FAMILY = ("gemm_nt_kernel", "wgrad_tn_kernel", "wgrad_tn_dma_kernel", "conv_patch_kernel")


def traced_kernels(st, extra_args=(), steps=20, warmup=5, timeout=600):
    """Per-kernel durations of the REPLAYED step: this command again as a child under `rocprofv3 --kernel-trace --stats`
    (graph replays have no per-launch hooks for HIP events; the profiler sees the replayed kernels themselves). Returns
    {kernel base name: (calls per step, avg us)} and the number of steps in the trace, or (None, reason). The child is a fresh
    process started as `rocprofv3 ... -- python3 bench.py ...` (never an exec from this GPU-initialised process)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    import re
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="cpcsv_trace_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--stats", "-d", work, "-o", "bench", "--", sys.executable, os.path.abspath(__file__),
           "--steps", str(steps), "--warmup", str(warmup), "--st", str(st), "--no-cpu-baseline", "--no-meter", "--child"] + list(extra_args)
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(work) for f in fs if f.endswith(".db")]
        if r.returncode != 0 or not dbs:
            return None, "rocprofv3 child failed (rc %d): %s" % (r.returncode, (r.stderr or "")[-300:])
        db = sqlite3.connect(dbs[0])
        cols = [d[1] for d in db.execute("pragma table_info(top_kernels)")]
        rows = [dict(zip(cols, x)) for x in db.execute("select * from top_kernels")]
        base0 = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n)).replace("void ", "")).strip()

        def base(n):
            b = base0(n)
            if b.startswith("thin"):       # the streaming convolutions: one kernel name serves several widths - keep the template arguments
                m = re.search(r"(thin\w+<[^>]*>)", n)
                return m.group(1).replace(" ", "") if m else b
            if b.startswith("at::"):       # torch's generic launchers: keep WHAT they apply (the functor), or nobody can tell a fill from an add
                m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w*(?:Functor|Op|functor)\w*(?:<[\w:]+>)?)", n[n.find("<"):] if "<" in n else "")
                if not m:
                    m = re.search(r"at::native::(\w+_cuda)\(", n)
                if m:
                    b += "<" + m.group(1) + ">"
            return b
        nsteps = max(1, sum(x["total_calls"] for x in rows if base(x["name"]) == "adam_kernel") // 4)
        out = {}
        for x in rows:
            b = base(x["name"])
            c, t = out.get(b, (0, 0.0))
            out[b] = (c + x["total_calls"], t + x["total_duration"])
        try:
            child = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            child = {}
        return {k: (c / nsteps, t / max(c, 1)) for k, (c, t) in out.items()}, {"steps": nsteps, "ms_per_step_under_rocprof": child.get("ms_per_step")}
    except Exception as e:                                     # pragma: no cover - depends on the box
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def child_ms_per_step(st, extra_args, steps=10, warmup=5, timeout=600):
    """ms/step of this command with other flags (the fp32 / reference-arithmetic mode), in a fresh child process."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--st", str(st),
           "--no-cpu-baseline", "--no-meter", "--child"] + list(extra_args)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        return json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:                                          # pragma: no cover
        return None


def cpu_baseline(st, im, timed=2, cascade=False, dims="pororo", threads=None):
    """The oracle (CPU fp32 restatement of trainer.py:252-416, pinned to the reference by tests/golden/) on this host's
    cores, on the bench workload itself: same widths, same ST/IM batch, same synthetic-batch seed; 1 warm-up step +
    `timed` timed steps. MKL-DNN does not scale to hundreds of threads on these layer sizes (measured:
    profiles/r05_cpu_threads.txt), so at most 32 threads are used unless `threads` says otherwise; the count is reported."""
    from oracle.cpcsv_oracle import clevr_cfg, make_state, pororo_cfg as ocfg, synthetic_batch, train_step
    cores = int(threads) if threads else min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    oc = (clevr_cfg if dims == "clevr" else ocfg)(st_batch=st, im_batch=im, cascade=cascade)
    state = make_state(oc, seed=0)
    stb, imb = synthetic_batch(oc, seed=1)
    train_step(state, stb, imb)                       # warm-up (allocator, MKL-DNN primitive caches)
    t0 = time.time()
    for _ in range(timed):
        train_step(state, stb, imb)
    dt = (time.time() - t0) / timed
    return {"value": round(st * oc.video_len / dt, 4), "unit": "story-frames/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "the bench workload itself (%sST=%d IM=%d, cfg/final.yml widths, fp32): 1 warm-up + %d timed steps, "
                      "%.1f s per step" % ("CLEVR dims T=4 / text 72 / 15 labels, " if dims == "clevr" else "", st, im, timed, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--meter-inline", action="store_true", help="HIP events around the GEMM launches INSIDE the timed region")
    ap.add_argument("--st", type=int, default=None, help="stories per rank (default: BASELINE config 2's 12; 2 with --clevr)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-meter", action="store_true")
    ap.add_argument("--cascade", action="store_true",
                    help="cascade_model.StoryGAN (CASCADE_MODEL: True; BASELINE config 4's generator at 64x64) instead of config 2's")
    ap.add_argument("--seq", action="store_true",
                    help="USE_SEQ_CONSISTENCY: the story critic also trains the VideoEncoder order critic (SURVEY F1; reference model.py:99-210)")
    ap.add_argument("--clevr", action="store_true",
                    help="BASELINE config 1's dimensions (CLEVR: T=4, text 72, 15 labels; ST=2 / IM=8 - batch 1 is impossible under "
                         "BatchNorm1d) at cfg/final.yml widths: the reference's own CPU-runnable case, with its CPU-port time beside it")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (default: min(cores, 32))")
    ap.add_argument("--child", action="store_true", help="(internal) this process is a measurement child of another bench.py: no children of its own")
    ap.add_argument("--no-trace", action="store_true", help="skip the rocprofv3 child that times the replayed kernels (roofline falls back to the eager meter)")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the fp32-mode child (fp32_ms_per_step)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as ONE captured HIP graph (trainer.train_step_graphed) instead of the default, twelve "
                         "capture-once graphs of its self-contained pieces on concurrent streams (cpcsv/graphs.py). One graph "
                         "serialises the branches more than the piecewise form does; kept as a tested option, not the fast path")
    args = ap.parse_args()
    dims = "clevr" if args.clevr else "pororo"
    if args.st is None:                          # (an explicit --st is honoured, also with --clevr)
        args.st = 2 if args.clevr else 12
    st, im = (args.st, 4 * args.st) if args.clevr else (args.st, 5 * args.st)

    from cpcsv import dist as cdist
    from cpcsv import runtime
    rank, world, local = cdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    runtime.set_compute_dtype(args.dtype)
    pororo_cfg(st, im, cascade=args.cascade, seq=args.seq, dims=dims)

    import trainer as T
    torch.manual_seed(0)                         # identical replicas (main_pororo.py:53)
    tr = T.GANTrainer(None, types.SimpleNamespace(cfg_file=None, continue_ckpt=None), ratio=1.0)
    tr.setup()
    torch.manual_seed(1000 + rank)               # per-rank noise stream
    st_batch, im_batch = synthetic_batches(st, im, 1 + rank, dev, dims=dims)

    meter = GemmMeter()
    if not args.no_meter:
        meter.install()

    def barrier():
        if world > 1 or cdist.is_distributed():
            cdist.barrier()          # (on the communication stream: cpcsv/dist.py _sync_collective)
        torch.cuda.synchronize()

    os.environ["CPCSV_GRAPH"] = "1" if args.graph else "0"
    step = lambda a, b: tr.train_step_graphed(a, b)
    # untimed: W warm-up steps (+ the eager steps / capture the graph path needs before it can replay)
    for _ in range(max(args.warmup, 5 if args.graph else 0)):
        stats = step(st_batch, im_batch)
    graphed = tr.__dict__.get("_gs", {}).get("graph") is not None
    barrier()
    inline = False   # (--meter-inline is accepted for compatibility; graph replays cannot be metered in place)
    if not args.no_meter:
        meter.calibrate()
    meter.on = inline
    # same host policy as GANTrainer.train(): the cyclic garbage collector is off inside the step loop (collections
    # are run between epochs / every 200 iterations there); a collection pause stalls the launch stream
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step(st_batch, im_batch)
    barrier()
    dt = time.perf_counter() - t0
    meter.on = False
    metered_steps = args.steps
    if not args.no_meter and not inline:
        # The timed region replays captured HIP graphs (cpcsv/graphs.py), which have no per-launch hooks, and two event
        # records around each of the ~390 GEMM launches of a step would cost the host ~8 ms per step: the per-kernel
        # durations therefore come from the SAME step run eagerly for a few more steps right after the timed region
        # (same kernels, same descriptors). rocprofv3 (profiles/) sees the replayed kernels themselves.
        metered_steps = min(args.steps, 10)
        from cpcsv import graphs
        graphs.PAUSED[0] = True          # same kernels, same descriptors, launched one by one so that the hooks see them
        for _ in range(2):
            tr.train_step(st_batch, im_batch)
        torch.cuda.synchronize()
        meter.on = True
        for _ in range(metered_steps):
            tr.train_step(st_batch, im_batch)
        torch.cuda.synchronize()
        meter.on = False
        graphs.PAUSED[0] = False
    gc.enable()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        cdist.all_reduce_max(tt)                  # (on the communication stream, like every collective of the step)
        dt = tt.item()
    loss = float(stats["G/loss"])
    if (not (loss == loss) or abs(loss) == float("inf")) and not os.environ.get("CPCSV_BENCH_ALLOW_NONFINITE"):   # (tools/ablate.sh)
        raise SystemExit("non-finite generator loss after the timed steps: %r" % loss)

    if rank == 0:
        frames = world * st * DIMS[dims][0] * args.steps
        line = {
            "metric": "story-frames/sec per train step (Pororo 64x64, seq_len=5)",
            "value": round(frames / dt, 2), "unit": "story-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (("clevr64_seq4" if args.clevr else "pororo64_seq5") + "_st%d_im%d_per_gpu_final_yml_widths"
                                    + ("_cascade" if args.cascade else "") + ("_order_critic" if args.seq else "")) % (st, im),
                       "global_story_batch": world * st, "global_image_batch": world * im,
                       "parallelism": "dp%d" % world, "G_loss_after": round(loss, 4),
                       "launch": ("hip_graph_replay" if graphed else
                                  "piecewise_hip_graphs" if getattr(tr.__dict__.get("_ng"), "captured", False) else "eager")},
        }
        if not args.no_meter:
            flops, executed, ms, n = meter.summary()
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            exe = executed / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            # achieved = ALGORITHMIC FLOP (the reference's 9-tap conv on the upsampled map) / time; executed_tflops =
            # what the MFMAs actually did (the sub-pixel form of upsample+conv needs 2.25x fewer)
            traffic, traffic_src = pmc_traffic()
            line["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                                "avg_launch_us": round(1e3 * ms / max(n, 1), 2),
                                "event_pair_overhead_us": round(1e3 * meter.overhead_ms, 2),
                                "kernel": "gemm_nt_kernel / wgrad_tn_kernel (MFMA gather-GEMM family)",
                                "executed_tflops": round(exe, 2), "executed_frac": round(exe / peak, 4),
                                "launches": n, "gemm_ms_per_step": round(ms / metered_steps, 3),
                                "gflop_per_step": round(flops / metered_steps / 1e9, 1),
                                "streaming_convs_hbm": meter.thin_summary(),
                                "timed_with": "HIP events on the launch stream, event-pair overhead subtracted, %s" % (
                                    "inside the timed region" if inline else
                                    "%d more steps of the same loop right after the timed region" % metered_steps)}
            # The same quantities from the REPLAYED step: a child of this command under rocprofv3 --kernel-trace (20 timed
            # steps). `frac` / `achieved` are then priced with the trace's durations - what the step really gets - and the eager
            # HIP-event figures are kept beside them (the eager launches run with less concurrency and read a little faster).
            if world == 1 and not args.child and not args.no_trace and args.dtype == "bf16":
                extra = (["--cascade"] if args.cascade else []) + (["--seq"] if args.seq else []) + (["--clevr"] if args.clevr else [])
                kern, info = traced_kernels(st, extra)
                rl = line["roofline"]
                if kern is None:
                    rl["trace"] = {"error": info}
                else:
                    fam_ms = sum(c * t for k, (c, t) in kern.items() if k in _GEMM_FAMILY) / 1e3
                    fam_n = sum(c for k, (c, t) in kern.items() if k in _GEMM_FAMILY)
                    gflop = flops / metered_steps / 1e9
                    rl["eager_events"] = {"achieved": rl["achieved"], "frac": rl["frac"], "gemm_ms_per_step": rl["gemm_ms_per_step"],
                                          "avg_launch_us": rl["avg_launch_us"], "executed_tflops": rl["executed_tflops"]}
                    ach = gflop / fam_ms if fam_ms > 0 else 0.0
                    rl.update({"achieved": round(ach, 2), "frac": round(ach / peak, 4), "gemm_ms_per_step": round(fam_ms, 3),
                               "avg_launch_us": round(1e3 * fam_ms / max(fam_n, 1), 2), "launches_per_step": round(fam_n, 1),
                               "executed_tflops": round(executed / metered_steps / 1e9 / fam_ms, 2) if fam_ms > 0 else 0.0,
                               "executed_frac": round(executed / metered_steps / 1e9 / fam_ms / peak, 4) if fam_ms > 0 else 0.0,
                               "timed_with": "rocprofv3 --kernel-trace of this command in a child process: %d steps, graph replays; "
                                             "algorithmic FLOP per step from the eager meter" % info["steps"],
                               "trace": info})
                    # streaming convolutions: algorithmic bytes per launch (meter) / the replayed kernels' average duration
                    names = {"thin3x3_fwd_c128": "thin3x3_fwd_taps_kernel", "thin3x3_fwd_c64": "thin3x3_fwd_roll_kernel",
                             "thin3x3_dgrad_c128": "thin3x3_dgrad_kernel", "thin3x3_dgrad_c64": "thin3x3_dgrad_kernel",
                             "thin3x3_wgrad_c128": "thin3x3_wgrad_rows_kernel", "thin3x3_wgrad_c64": "thin3x3_wgrad_rows_kernel",
                             "thin4x4s2_fwd_c128": "thin4x4s2_fwd_kernel", "thin4x4s2_dgrad_c128": "thin4x4s2_dgrad_kernel",
                             "thin4x4s2_wgrad_c128": "thin4x4s2_wgrad_kernel"}
                    for key, row in rl["streaming_convs_hbm"].items():
                        kn = names.get(key)
                        width = key.rsplit("_c", 1)[-1]
                        cands = [k for k in kern if kn and (k == kn or k.startswith(kn + "<"))]
                        if len(cands) > 1:                  # one kernel name, several widths: the instantiation of THIS row's width
                            cands = [k for k in cands if k.startswith("%s<%s" % (kn, width))]
                        if len(cands) == 1:
                            us = kern[cands[0]][1]
                            row["replayed_avg_us"] = round(us, 1)
                            row["replayed_TB_per_s"] = round(row["MB_per_launch"] / us, 3)                # MB / us = TB/s
                            row["replayed_frac_of_8TBps"] = round(row["replayed_TB_per_s"] / 8.0, 3)
                    line["kernel_ms_per_step"] = round(sum(c * t for c, t in kern.values()) / 1e3, 3)
                    line["launches_per_step"] = round(sum(c for c, t in kern.values()), 1)
                    line["non_cpcsv_launches_per_step"] = round(sum(c for k, (c, t) in kern.items() if k.startswith(("at::", "__amd_rocclr"))), 1)
                    import re as _re
                    short = lambda k: _re.sub(r"at::native::|\(anonymous namespace\)::|<unnamed>::", "", k)[:72]
                    line["non_cpcsv_kernels"] = {short(k): round(c, 1) for k, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][0])
                                                 if k.startswith(("at::", "__amd_rocclr"))}
            if os.environ.get("CPCSV_BENCH_SHAPES"):
                with open(os.environ["CPCSV_BENCH_SHAPES"], "w") as fh:
                    fh.write("\n".join(meter.by_shape()) + "\n")
        if world == 1 and not args.child and not args.no_fp32_line and args.dtype == "bf16":
            # the reference's own arithmetic (fp32, exact f32 MFMA) on the same workload, fresh process
            line["fp32_ms_per_step"] = child_ms_per_step(st, ["--dtype", "fp32"] + (["--cascade"] if args.cascade else []) + (["--seq"] if args.seq else [])
                                                         + (["--clevr"] if args.clevr else []))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(st, im, cascade=args.cascade, dims=dims, threads=args.cpu_threads or None)
        print(json.dumps(line), flush=True)
    cdist.shutdown()          # drain the device, barrier, drain, destroy (cpcsv/dist.py)


if __name__ == "__main__":
    main()
