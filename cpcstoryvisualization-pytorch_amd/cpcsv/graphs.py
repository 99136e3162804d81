"""Capture-once / replay wrappers for the self-contained pieces of the training step.

A step is ~1900 kernel launches issued by one Python thread; the pieces that have fixed shapes and no data-dependent
host control flow (the no-grad generator pass, each critic's forward+backward) are captured ONCE as HIP graphs after
a few eager calls and replayed, so their launches no longer queue up behind the interpreter. Everything that changes
from step to step lives on the device (weights, Adam counters, spectral-norm u/v, BatchNorm statistics, RNG offsets
through torch's graph-safe generator); the little host-side bookkeeping the eager code does is redone after a replay.
"""
import os

import torch

from . import modules as M
from . import runtime


PAUSED = [False]       # bench.py's per-kernel metering pass runs the same step eagerly (a replay has no launch hooks)


class GraphedCall:
    """fn(*tensors) -> tensors (or a dict of tensors), captured on `stream` (None: torch's capture side stream).

    * the first `warmup` calls run eagerly (lazy buffers, descriptors, split-K workspaces, Adam tables);
    * inputs are copied into static buffers before a replay, outputs are the graph's static tensors (valid until the
      next replay);
    * BatchNorm call counters of `bn_owner`'s modules advance per replay like they do per eager call;
    * layers whose packed weights are rebuilt INSIDE the captured region get their pack keys moved forward after a
      replay (the replay rebuilt them from the current weights);
    * falls back to eager for good if the runtime refuses the capture, and per call if `enabled()` is false, the
      shapes changed or another capture is in progress."""

    def __init__(self, fn, name, bn_owner=None, stream=None, warmup=None, enabled=lambda: True, pool_from=None):
        if warmup is None:
            warmup = int(os.environ.get("CPCSV_GRAPH_WARMUP", "3"))      # eager calls before the capture (>= 1)
        self.fn, self.name, self.bn_owner, self.stream, self.warmup, self.enabled = fn, name, bn_owner, stream, max(1, warmup), enabled
        self.calls, self.graph, self.off, self.terms, self.updates = 0, None, False, [], []
        self.pool_from = pool_from      # another GraphedCall whose autograd graph this one's backward walks into: one pool

    def _capture(self, ins):
        bns = [m for m in (self.bn_owner.modules() if self.bn_owner is not None else []) if hasattr(m, "note_batch")]
        before = [m._pending for m in bns]
        static = tuple(t.clone() for t in ins)
        M.PACK_LOG, M.USE_LOG, M.TERM_LOG, M.UPDATE_LOG = [], [], [], []
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            kw = {"stream": self.stream} if self.stream is not None else {}
            kw["capture_error_mode"] = "thread_local"     # e.g. the RCCL watchdog thread may touch HIP meanwhile
            if self.pool_from is not None and self.pool_from.captured:
                kw["pool"] = self.pool_from.graph.pool()
            with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(g, **kw):   # backward on this thread
                outs = self.fn(*static)
            self.graph, self.outs, self.static, self.packs, self.terms = g, outs, static, M.PACK_LOG, M.TERM_LOG
            _drop_capture_time_terms(self.terms)
            self.updates = list(M.UPDATE_LOG)
            self._note_uses(M.USE_LOG)
            self.bn = [(m, m._pending - b) for m, b in zip(bns, before)]
            for m, b in zip(bns, before):
                m._pending = b                      # the capture executed nothing
            return True
        except Exception as e:                   # pragma: no cover - depends on the runtime
            self.off = True
            print("[cpcsv] HIP graph capture of %s refused (%s: %s); staying eager" % (self.name, type(e).__name__, e))
            torch.cuda.synchronize()
            return False
        finally:
            M.PACK_LOG = M.USE_LOG = M.TERM_LOG = M.UPDATE_LOG = None

    def _note_uses(self, uses):
        """Operand sets the captured kernels READ but do not rebuild themselves: they rely on an earlier replay of
        another piece having rewritten those buffers in place from the current weights."""
        inside = {(id(layer), part) for layer, _, _, parts in self.packs for part in parts}
        seen, self.reads = set(), []
        for layer, weight, dt, parts in uses:
            for part in parts:
                k = (id(layer), part)
                if k not in inside and k not in seen:
                    seen.add(k)
                    self.reads.append((layer, weight, dt, part))

    def _refresh_stale(self):
        """A weight changed outside the captured flow (load_state_dict, a manual edit, an optimiser step with no
        repacking piece after it): rebuild the operand copies this replay is about to read, eagerly, in place, on the
        stream the graph runs on. In the steady state of the training loop nothing is stale and this is a key compare."""
        for layer, weight, dt, part in self.reads:
            if layer._key.get(part) != layer.pack_key(weight, dt):
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        layer.packs(weight, dt, part)
                else:
                    layer.packs(weight, dt, part)

    def __call__(self, *ins):
        if self.off or PAUSED[0] or not self.enabled() or torch.cuda.is_current_stream_capturing():
            return self.fn(*ins)
        if self.calls < self.warmup:
            self.calls += 1
            return self.fn(*ins)
        if self.graph is None and not self._capture(ins):
            return self.fn(*ins)
        if any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(ins, self.static)):
            return self.fn(*ins)
        _copy_inputs(self.static, ins)
        self._refresh_stale()
        self.graph.replay()
        for m, k in self.bn:
            m._pending += k
        for layer, weight, dt, parts in self.packs:
            layer.mark_packed(weight, dt, parts)
        for layer, term in self.terms:                  # spectral-norm terms the replayed backward left for the deferred update
            layer.fused_terms.append(term)
        for layer in self.updates:                      # ... unless the replay applied that layer's update itself
            layer.fused_updated = True
        return self.outs

    @property
    def captured(self):
        return self.graph is not None


def _copy_inputs(static, ins):
    """This call's input tensors into the graph's static buffers: ONE launch for all contiguous same-dtype pairs
    (cpcsv_copy_many) instead of a copy launch per tensor; anything else (strided views, dtype casts) through torch."""
    pairs = []
    for dst, src in zip(static, ins):
        if dst.data_ptr() == src.data_ptr():
            continue
        if src.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype and src.shape == dst.shape:
            pairs.append((dst, src))
        else:
            dst.copy_(src, non_blocking=True)
    if pairs:
        from . import kernels as K
        K.copy_many(pairs)


def _drop_capture_time_terms(terms):
    """The Python of a captured backward ran once during the capture and queued its spectral-norm terms on the layers,
    but none of its kernels executed: un-queue them (every replay re-queues them, see GraphedCall.__call__)."""
    for layer, term in terms:
        for i, t in enumerate(layer.fused_terms):
            if t is term:
                del layer.fused_terms[i]
                break


def _flatten(obj, out):
    """nested tuples/lists of tensors/None/constants -> spec; tensors appended to `out`"""
    if torch.is_tensor(obj):
        out.append(obj)
        return ("t", len(out) - 1)
    if isinstance(obj, (tuple, list)):
        return ("s", type(obj), [_flatten(o, out) for o in obj])
    return ("c", obj)


def _unflatten(spec, tensors):
    if spec[0] == "t":
        return tensors[spec[1]]
    if spec[0] == "s":
        return spec[1](_unflatten(s, tensors) for s in spec[2])
    return spec[1]


class _Replay(torch.autograd.Function):
    """forward = replay of the captured forward graph, backward = replay of the captured backward graph."""

    @staticmethod
    def forward(ctx, owner, hook, *ins):
        _copy_inputs(owner.static, ins)
        owner._refresh_stale()
        owner.fwd_graph.replay()
        owner._after_replay()
        ctx.owner = owner
        outs = tuple(o.detach() for o in owner.flat_outs)
        ctx.mark_non_differentiable(*[o for o, rg in zip(outs, owner.out_rg) if not rg])
        ctx.set_materialize_grads(False)        # backward() handles None; a materialised zero is a fill launch per output
        return outs

    @staticmethod
    def backward(ctx, *grads):
        owner = ctx.owner
        it = iter(owner.static_grads)
        pairs = []
        for g, rg in zip(grads, owner.out_rg):
            if rg:
                sg = next(it)
                if g is None:
                    if not getattr(sg, "_cpcsv_is_zero", False):     # (a fill launch per replay otherwise: the static gradient of an
                        sg.zero_()                                   #  unused output stays zero from one step to the next)
                        sg._cpcsv_is_zero = True
                    continue
                sg._cpcsv_is_zero = False
                if g.is_contiguous() and sg.is_contiguous() and g.dtype == sg.dtype and g.shape == sg.shape:
                    if g.data_ptr() != sg.data_ptr():
                        pairs.append((sg, g))
                else:
                    sg.copy_(g, non_blocking=True)
        if pairs:
            from . import kernels as K
            K.copy_many(pairs)
        owner.bwd_graph.replay()
        owner._after_backward_replay()
        gin = [None] * len(owner.static)
        for i, g in zip(owner.grad_inputs, owner.static_gin):
            gin[i] = g
        return (None, None) + tuple(gin)


class GraphedAutograd(GraphedCall):
    """Like GraphedCall for a DIFFERENTIABLE piece: fn(*tensors) builds an autograd graph; its forward and its
    backward are captured as two HIP graphs sharing one memory pool (the scheme of torch.cuda.make_graphed_callables)
    and stitched into the surrounding eager autograd graph by one Function. Parameter gradients are not returned: the
    layers accumulate them in place into the persistent flat gradient buffers (dist.GradBucket.adopt), which the
    backward graph does as a side effect; `grad_inputs` lists the inputs whose gradient the caller needs."""

    def __init__(self, fn, name, bn_owner=None, stream=None, warmup=None, enabled=lambda: True, grad_inputs=(), wgrad_stream=None,
                 late_stream=None):
        super().__init__(fn, name, bn_owner, stream, warmup, enabled)
        self.grad_inputs = tuple(grad_inputs)
        self.wgrad_stream = wgrad_stream       # parallel branch of the backward graph for the weight gradients
        self.late_stream = late_stream         # third branch: parked per-layer optimiser launches (runtime.set_late_stream)
        self.hook = None
        self.capturing = False

    def _after_replay(self):
        for m, k in self.bn:
            m._pending += k
        for layer, weight, dt, parts in self.packs:
            layer.mark_packed(weight, dt, parts)

    def _after_backward_replay(self):
        for layer, term in self.terms:
            layer.fused_terms.append(term)
        for layer in self.updates:
            layer.fused_updated = True

    def _capture(self, ins):
        bns = [m for m in (self.bn_owner.modules() if self.bn_owner is not None else []) if hasattr(m, "note_batch")]
        before = [m._pending for m in bns]
        static = tuple(t.detach().clone().requires_grad_(i in self.grad_inputs) for i, t in enumerate(ins))
        M.PACK_LOG, M.USE_LOG, M.TERM_LOG, M.UPDATE_LOG = [], [], [], []
        self.capturing = True
        try:
            torch.cuda.synchronize()
            kw = {"stream": self.stream} if self.stream is not None else {}
            kw["capture_error_mode"] = "thread_local"
            gf = torch.cuda.CUDAGraph()
            mt = torch.autograd.set_multithreading_enabled(False)      # the captured backward runs on this thread
            mt.__enter__()
            with torch.cuda.graph(gf, **kw):
                outs = self.fn(*static)
            flat = []
            spec = _flatten(outs, flat)
            out_rg = [o.requires_grad for o in flat]
            rg_outs = [o for o in flat if o.requires_grad]
            static_grads = [torch.zeros_like(o) for o in rg_outs]
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=gf.pool(), **kw):
                if self.grad_inputs:
                    gin = torch.autograd.grad(rg_outs, [static[i] for i in self.grad_inputs], grad_outputs=static_grads,
                                              allow_unused=True)
                else:
                    if self.wgrad_stream is not None:
                        runtime.set_wgrad_stream(self.wgrad_stream)
                        runtime.set_late_stream(self.late_stream)
                    try:
                        runtime.defer_small_wgrads(True)            # the small dense layers' weight gradients: ONE launch at the end
                        torch.autograd.backward(rg_outs, grad_tensors=static_grads)
                        runtime.flush_late(self.wgrad_stream)       # (no layer gave the signal: behind the whole chain)
                    finally:
                        runtime.defer_small_wgrads(False)
                        runtime.set_wgrad_stream(None)
                        runtime.set_late_stream(None)
                    if self.wgrad_stream is not None:
                        torch.cuda.current_stream().wait_stream(self.wgrad_stream)    # join the branches inside the capture
                    # the parked small weight gradients, behind the join: a > 64-row pass of the same layer may have written the
                    # same .grad on the weight-gradient branch (non-atomic read-modify-write on both sides)
                    runtime.flush_small_wgrads()
                    if self.wgrad_stream is not None:
                        if self.late_stream is not None:
                            torch.cuda.current_stream().wait_stream(self.late_stream)
                        runtime.release_kept()
                    gin = ()
            mt.__exit__(None, None, None)
            self.fwd_graph, self.bwd_graph, self.graph = gf, gb, gf
            self.static, self.flat_outs, self.spec, self.out_rg = static, flat, spec, out_rg
            self.static_grads, self.static_gin, self.packs, self.terms = static_grads, gin, M.PACK_LOG, M.TERM_LOG
            _drop_capture_time_terms(self.terms)
            self.updates = list(M.UPDATE_LOG)
            self._note_uses(M.USE_LOG)
            self.bn = [(m, m._pending - b) for m, b in zip(bns, before)]
            for m, b in zip(bns, before):
                m._pending = b
            self.hook = torch.zeros(1, device=static[0].device, requires_grad=True)   # makes the outputs require grad
            return True
        except Exception as e:                   # pragma: no cover - depends on the runtime
            self.off = True
            print("[cpcsv] HIP graph capture of %s refused (%s: %s); staying eager" % (self.name, type(e).__name__, e))
            torch.cuda.synchronize()
            return False
        finally:
            M.PACK_LOG = M.USE_LOG = M.TERM_LOG = M.UPDATE_LOG = None
            self.capturing = False

    def __call__(self, *ins):
        if self.off or PAUSED[0] or not self.enabled() or torch.cuda.is_current_stream_capturing() or not torch.is_grad_enabled():
            return self.fn(*ins)
        if self.calls < self.warmup:
            self.calls += 1
            return self.fn(*ins)
        if self.graph is None and not self._capture(ins):
            return self.fn(*ins)
        if any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(ins, self.static)):
            return self.fn(*ins)
        outs = _Replay.apply(self, self.hook, *ins)
        return _unflatten(self.spec, outs)


def env_on(name, default="1"):
    return os.environ.get(name, default) != "0"


def many_graphs_safe():
    """ROCm 7.2's graph "packet capture" fast path corrupts earlier executable graphs once the live graphs of a
    process hold more than ~2900 kernel nodes in total (measured here: the critics' gradients turn into 1e14-1e40
    garbage a step or two after the generator's graphs are instantiated). It is switched off with
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, which the HIP runtime reads when it initialises: the entry points (bench.py,
    __graft_entry__.py, tests/conftest.py; a deployment: INTEGRATION.md) set it before torch is imported and cpcsv/runtime.py
    checks that they did (PACKET_CAPTURE_OFF). Only then are ALL pieces of the step captured; otherwise the no-grad pass and
    the critic updates (2400 nodes, tested) are."""
    return runtime.PACKET_CAPTURE_OFF or os.environ.get("CPCSV_MANY_GRAPHS") == "1"      # (=1: experiments only)
