"""Data-parallel gradient exchange: one process per GPU, RCCL (backend 'nccl') over xGMI.

The reference has no torch.distributed at all (single-process nn.parallel.data_parallel on the
critics only, miscc/utils.py:58-166). Stock DistributedDataParallel cannot wrap these
nets (StoryGAN has no forward(); critic heads are called outside netD.forward), so the exchange is
explicit: all gradients of one optimiser are flattened into ONE buffer and mean-all-reduced right
before that optimiser's step. Few, large collectives are what point-to-point xGMI wants: RCCL
splits one big all-reduce across all 7 links (reduce-scatter + all-gather), a per-tensor ring would
be latency-bound. BatchNorm stays per-rank (not SyncBN): each rank reproduces the single-GPU step on
its own shard. Works unchanged with backend 'gloo' on CPU tensors (tests).
"""
import os

import torch
import torch.distributed as dist


_STORE_FIRST = os.environ.get("CPCSV_STORE_FIRST", "1") != "0"
_POISON_SKIPPED = os.environ.get("CPCSV_POISON", "0") == "1"      # debugging / tests: NaN into every accumulator whose fill is skipped


def force_exchange():
    """CPCSV_FORCE_EXCHANGE=1: run the whole data-parallel machinery - process group, chunked asynchronous all-reduces between
    the graph pieces, optimiser steps after the exchange - even with ONE rank. A rehearsal of the RCCL path for boxes with a
    single GPU (tests/test_gpu_dist.py::test_rccl_world1_rehearsal, tools/rccl_rehearsal.sh): same results, measurable cost."""
    return os.environ.get("CPCSV_FORCE_EXCHANGE", "0") == "1"


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_exchange())


def init_from_env(backend=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (torchrun). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("CPCSV_FORCE_DEVICE") or os.environ.get("LOCAL_RANK", "0"))   # FORCE_DEVICE: test aid, see below
    if (world > 1 or force_exchange()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:       # two jobs on one host must not meet on a silent default port
                raise RuntimeError("WORLD_SIZE=%d but MASTER_PORT is not set (torchrun / the launcher sets it)" % world)
            os.environ["MASTER_PORT"] = "29577"      # world-1 rehearsal (CPCSV_FORCE_EXCHANGE=1): a private rendezvous
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:     # CPCSV_DIST_BACKEND=gloo: several ranks on ONE GPU (single-GPU test boxes); RCCL refuses that
            backend = os.environ.get("CPCSV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


# ---- ONE communication stream ----------------------------------------------------------------------------------------------
# Every collective of this package - the synchronous ones (flat-buffer all-reduce, broadcast, barrier) and the chunked exchange of
# the layer accumulators - is issued on ONE dedicated HIP stream, in host order:
#   * one communicator + one stream = the only configuration NCCL / RCCL guarantees without further assumptions: the collectives
#     execute in issue order, which is the same program order on every rank (rounds 4-5 issued synchronous collectives from the three
#     critic streams onto one communicator, beside asynchronous ones on ProcessGroupNCCL's internal stream: never run on > 1 rank);
#   * no collective's end event is ever recorded on a stream that is captured into a HIP graph. torch >= 2.7 runs a synchronous
#     collective on the CURRENT stream and records the work's end event there; the ProcessGroupNCCL watchdog polls un-retired works
#     every 100 ms, and hipEventQuery on an event last recorded in a now-capturing stream is an error that invalidates the capture
#     and aborts the process from the watchdog thread (the SIGABRT of round 4's 8-GPU run; profiles/r05_rccl_soak.txt). Round 5 kept
#     collectives off the capture streams only while captures were expected and guarded late captures with a 350 ms sleep; here the
#     critic streams and the main stream never see a collective at all, so there is nothing to wait out.
# The stream is created and submitted to in GANTrainer.__init__ (bind_comm_stream): the HIP runtime binds a stream to one of its 4
# hardware queues at the stream's first submission, round robin, and two streams on one queue run strictly one after the other
#  - which queue the communication stream shares is therefore decided there, not left to the first collective.
_COMM = {"stream": None}


def comm_stream():
    if _COMM["stream"] is None:
        _COMM["stream"] = torch.cuda.Stream()
    return _COMM["stream"]


def bind_comm_stream(spacers=0):
    """First submission of the communication stream (after `spacers` throw-away streams took the queues in front of it)."""
    keep = []
    t = torch.zeros(8, device=torch.device("cuda", torch.cuda.current_device()))
    for _ in range(spacers):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            t.add_(1.0)
        keep.append(s)
    with torch.cuda.stream(comm_stream()):
        t.add_(1.0)
    torch.cuda.synchronize()
    return keep


def _on_comm_stream(tensor, group=None):
    return tensor.is_cuda and dist.get_backend(group) == "nccl"


def _sync_collective(launch, tensor, group=None):
    """A synchronous collective on a device tensor: issued on the communication stream behind everything the caller's stream has
    enqueued, and the caller's stream continues behind it. launch(async_op) issues it."""
    if not _on_comm_stream(tensor, group):
        launch(False)
        return
    cur, cs = torch.cuda.current_stream(), comm_stream()
    if cur.cuda_stream == cs.cuda_stream:
        launch(False)
        return
    cs.wait_stream(cur)
    with torch.cuda.stream(cs):
        launch(False)            # async_op=False: runs on the CURRENT stream = cs, end event recorded there
    cur.wait_stream(cs)


def barrier():
    """dist.barrier() that leaves no event behind on a stream that may capture later (see _sync_collective)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if torch.cuda.is_available() and dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        _sync_collective(lambda a: dist.barrier(device_ids=[dev.index], async_op=a), torch.empty(0, device=dev))
    else:
        dist.barrier()


def all_reduce_max(tensor, group=None):
    """In-place MAX all-reduce of a (small) tensor through the same path as every other collective of the package (bench.py's
    max-over-ranks time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    _sync_collective(lambda a: dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=group, async_op=a), tensor, group)
    return tensor


def shutdown():
    """Leave the process group with nothing in flight: drain the device, barrier, drain again, destroy. (Round 4 blamed its
    1-in-14 SIGABRT on the order barrier -> destroy; the abort is the watchdog's captured-event query described at
    _sync_collective and strikes during the capture steps, whatever the teardown does - tools/rccl_soak.sh OLD=1 / OLD=0: 3 of 15
    and 3 of 30 before that fix.)"""
    if not (dist.is_available() and dist.is_initialized()):
        return
    cuda = torch.cuda.is_available() and dist.get_backend() == "nccl"
    if cuda:
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
    else:
        dist.barrier()
    dist.destroy_process_group()


class GradBucket:
    """All gradients of one optimiser in ONE flat fp32 buffer.

    `adopt()` makes every parameter's .grad a view into the flat buffer and marks the parameters so the
    HIP backward kernels accumulate into it directly (cpcsv.functional.LayerFn): zeroing is one memset,
    the all-reduce is one collective on the buffer itself, and the pointers Adam's table holds never change."""

    def __init__(self, params, payload=None):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.adopted = False
        # what travels over xGMI: "fp32" (exact mean) or "bf16" (half the bytes: 316 instead of 632 MB per step for the
        # four nets; the sum is still formed in fp32 by RCCL's reduction of bf16 inputs only to bf16 precision, so this
        # is the bf16-training option, never used in fp32 parity mode). CPCSV_GRAD_COMM overrides.
        self.payload = os.environ.get("CPCSV_GRAD_COMM") or payload or "fp32"
        self._wire = None
        self.extra = []        # more flat fp32 gradient storage of the same optimiser (weight-gradient accumulators of the
        #                        deferred-update layers, cpcsv.optim.FusedAdam.attach_layer): zeroed and reduced with `flat`

    def adopt(self, retired=(), scalars=0):
        """`retired`: parameters whose gradient never materialises in master layout (the deferred-update weights: their
        gradient lives in the layer accumulators of `extra`). They keep a .grad view - behind the live part of the buffer,
        so that zeroing and the all-reduce, which work on `self.flat`, skip them (158 M of the 159 M elements at
        cfg/final.yml widths)."""
        dev = self.params[0].device
        dead = {id(p) for p in retired}
        self.params = [p for p in self.params if id(p) not in dead] + [p for p in self.params if id(p) in dead]
        live = sum(p.numel() for p in self.params if id(p) not in dead)
        # `scalars` extra floats behind the parameters' gradients, zeroed and mean-reduced with them: per-call <G, W> values of the
        # spectral-normed deferred-update layers (the rank-1 term of dL/dW_orig is linear in them and sigma, u, v are the same on
        # every rank, so the mean of the ranks' values is what the update after the exchange needs)
        self._storage = torch.zeros(live + scalars, dtype=torch.float32, device=dev)
        self.flat = self._storage
        self.scalars = self._storage[live:]
        # retired weights: no master-layout gradient exists (FusedAdam.export_grad rebuilds one on demand). Their .grad is
        # a stride-0 view of ONE zero - it keeps `p.grad is not None` (the kernels' "accumulate in place" marker) without
        # 632 MB of never-written storage at cfg/final.yml widths, and nothing may write through it.
        self._dummy = torch.zeros(1, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if id(p) in dead:
                p.grad = self._dummy.expand(p.shape)
                p._cpcsv_retired = True
            else:
                p.grad = self._storage[off:off + n].view_as(p)
                off += n
            p._cpcsv_direct = True
        self.adopted = True
        return self

    def zero(self):
        """Replacement for module.zero_grad() that keeps the persistent buffers. Accumulators of deferred-update layers whose first
        weight-gradient launch of a step STORES (one pixel slice: no float atomics into the buffer; cpcsv.functional.LayerFn learns it
        per layer, `store_first`) are not filled: 0.5 of the 0.62 GB a step's fills wrote at the benchmark's widths. The layer's
        backward fills on the spot if it ever finds it needs zeros after all (`_g_zeroed`). CPCSV_STORE_FIRST=0: fill everything."""
        if self.adopted:
            from . import kernels as K
            skip = {}
            if _STORE_FIRST:
                for lay in self.__dict__.get("fused_layers", ()):
                    sf = bool(getattr(lay, "store_first", False))
                    lay._g_zeroed = not sf
                    if sf:
                        skip.setdefault(lay._g_span[0], []).append(lay._g_span[1:])
            for bi, t in enumerate([self.flat] + self.extra):
                spans = sorted(skip.get(bi - 1, ())) if bi > 0 else ()
                if not t.is_cuda:
                    t.zero_()
                elif not spans:
                    K.fill_zero(t)
                else:
                    lo = 0                                   # the gaps between the skipped spans, merged
                    for a, b in spans + [(t.numel(), t.numel())]:
                        if a > lo:
                            K.fill_zero(t[lo:a])
                        if _POISON_SKIPPED and b > a:
                            t[a:b].fill_(float("nan"))       # (an element the layer's first launch does not overwrite surfaces as NaN)
                        lo = max(lo, b)
        else:
            for p in self.params:
                p.grad = None

    def norm(self):
        """L2 norm over everything this bucket holds (diagnostics/tests)."""
        return float(torch.sqrt(sum((t.double() ** 2).sum() for t in [self.flat] + self.extra)))

    def wire_of(self, bi):
        """bf16 wire buffer of accumulator buffer `bi` (payload "bf16"), else None."""
        return self.__dict__.get("_wires", {}).get(bi) if self.payload == "bf16" else None

    def reduce_extra_async(self, group=None, chunk_elems=None):
        """SUM all-reduce of the layer accumulators (`extra`) in chunks, issued back to back on the communication stream; returns
        [(buffer index, lo, hi, wait)] in issue order. `wait()` makes the CURRENT stream wait for that chunk. The caller
        applies 1/world (cpcsv_update_desc.gscale) and can start a chunk's fused layer updates while later chunks are still
        on the wire (cpcsv.optim.FusedAdam.step(pending=...)): over xGMI the generator's 348 MB take ~1.7 ms, its updates
        ~1 ms. Payload "bf16": each accumulator buffer is cast into its bf16 wire buffer (no pre-scaling: bf16 has fp32's exponent
        range), the chunks of the WIRE buffer are reduced, and the layer updates read their
        accumulator values from it (cpcsv_update_desc.g_bf16; wire_of()): half the bytes on the links, no conversion back; the
        cast runs chunk by chunk on the caller's stream, a chunk's all-reduce waits for its own cast only.
        Returns None when not distributed."""
        if not is_distributed() or not self.adopted or not self.extra:
            return None
        wire = self.payload == "bf16"
        if chunk_elems is None:
            chunk_elems = int(os.environ.get("CPCSV_COMM_CHUNK_MB", "64")) * (1 << 20) // (2 if wire else 4)
        out = []
        gloo_gpu = dist.get_backend(group) == "gloo"
        cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
        for bi, t in enumerate(self.extra):
            n = t.numel()
            src = t
            if wire:
                wires = self.__dict__.setdefault("_wires", {})
                src = wires.get(bi)
                if src is None or src.numel() != n or src.device != t.device:
                    src = wires[bi] = torch.empty(n, dtype=torch.bfloat16, device=t.device)
                if not t.is_cuda:
                    src.copy_(t)
            on_comm = _on_comm_stream(src, group)
            if on_comm and not wire:
                comm_stream().wait_stream(cur)                   # behind the backward pass that filled it
            for lo in range(0, n, chunk_elems):
                hi = min(n, lo + chunk_elems)
                part = src[lo:hi]
                if wire and t.is_cuda:
                    # fp32 -> bf16 chunk by chunk on the CALLER's stream; the communication stream waits for each chunk's cast only,
                    # so the casts of later chunks run beside the transfers of earlier ones
                    from . import kernels as K
                    K.copy2d(t[lo:hi], hi - lo, 0, part, hi - lo, 0, 1, hi - lo)
                    if on_comm:
                        ev_c = torch.cuda.Event()
                        ev_c.record(cur)
                        comm_stream().wait_event(ev_c)
                if gloo_gpu and part.is_cuda:      # test aid (several ranks on ONE GPU): host staging, synchronous
                    host = part.float().cpu() if wire else part.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                    part.copy_(host)
                    out.append((bi, lo, hi, lambda: None))
                elif on_comm:
                    with torch.cuda.stream(comm_stream()):
                        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
                        ev = torch.cuda.Event()
                        ev.record()
                    out.append((bi, lo, hi, lambda ev=ev: torch.cuda.current_stream().wait_event(ev)))
                else:
                    work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group, async_op=True)
                    out.append((bi, lo, hi, work.wait))
        return out

    def allreduce_mean(self, group=None, skip_extra=False):
        """In-place mean of .grad across ranks; no-op when not distributed. skip_extra: the layer accumulators are exchanged
        separately (reduce_extra_async)."""
        if not is_distributed():
            return
        world = dist.get_world_size(group)
        if self.adopted and self.extra and not skip_extra:
            keep_flat, keep_wire, keep_extra = self.flat, self._wire, self.extra
            try:                                  # same path for every extra buffer (own wire buffer each)
                self.extra = []
                for i, t in enumerate(keep_extra):
                    self.flat, self._wire = t, self.__dict__.setdefault("_wires", {}).get(i)
                    self.allreduce_mean(group)
                    self._wires[i] = self._wire
            finally:
                self.flat, self._wire, self.extra = keep_flat, keep_wire, keep_extra
        if self.adopted:
            if self.flat.is_cuda and dist.get_backend(group) == "gloo":
                # test aid (several ranks on ONE GPU, which RCCL refuses): gloo reduces host memory
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                self.flat.copy_(host.div_(world))
                return
            if self.payload == "bf16":
                if self._wire is None or self._wire.device != self.flat.device:
                    self._wire = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
                # pre-scale by 1/world so the bf16 sum cannot overflow and the mean needs no second pass
                if self._wire.numel() != self.flat.numel():
                    self._wire = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
                torch.mul(self.flat, 1.0 / world, out=self.flat)
                self._wire.copy_(self.flat)
                _sync_collective(lambda a: dist.all_reduce(self._wire, op=dist.ReduceOp.SUM, group=group, async_op=a), self._wire, group)
                self.flat.copy_(self._wire)
                return
            _sync_collective(lambda a: dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=a), self.flat, group)
            self.flat.div_(world)
            return
        plist = [p for p in self.params if p.grad is not None]
        if not plist:
            return
        n = sum(p.numel() for p in plist)
        dev = plist[0].grad.device
        if self.flat is None or self.flat.numel() != n or self.flat.device != dev:
            self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        views = []
        for p in plist:
            k = p.numel()
            v = self.flat[off:off + k]
            v.copy_(p.grad.reshape(-1))
            views.append(v)
            off += k
        _sync_collective(lambda a: dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=a), self.flat, group)
        self.flat.div_(world)
        for p, v in zip(plist, views):
            p.grad.copy_(v.view_as(p.grad))


def broadcast_module(module, src=0):
    """Make every rank start from rank `src`'s weights and buffers."""
    if not is_distributed():
        return
    gloo_gpu = dist.get_backend() == "gloo"
    for t in list(module.parameters()) + list(module.buffers()):
        if gloo_gpu and t.is_cuda:
            host = t.data.cpu()
            dist.broadcast(host, src)
            t.data.copy_(host)
        else:
            _sync_collective(lambda a, t=t: dist.broadcast(t.data, src, async_op=a), t.data)
