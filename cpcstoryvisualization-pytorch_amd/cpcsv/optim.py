"""Multi-tensor Adam on the HIP path (torch.optim.Adam semantics, reference trainer.py:212-220).

One kernel launch updates every parameter of an optimiser: a device-side pointer table
{p, g, m, v} per tensor plus a chunk map (block -> tensor, offset). torch's per-tensor foreach
path would be hundreds of launches for G's ~110 tensors; the reference's Adam is 1.8 % of its CPU step
and 4.4 GB/step of pure HBM streaming on the GPU, so it is one streaming pass here.
"""
import ctypes as C

import torch

from . import _lib as L
from . import kernels as K
from .runtime import dcode


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}
        self._hypers = {}
        self._layers = []          # deferred-update layers: [layer, weight, desc or None]
        self._fused_ids = set()
        self._ent_of = {}          # id(layer) -> entry
        self._hyper_next = None    # {step count of the COMING step, lr}: what in-backward layer updates read
        self._prepared = False
        import os
        self.inline = os.environ.get("CPCSV_INLINE_UPDATE", "1") != "0"     # the trainer clears it when world > 1

    def _table(self, gi, plist):
        """Device tables for one param group, rebuilt only when a pointer changed."""
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in plist)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1]
        dev = plist[0].device
        chunk = K.adam_chunk()
        ptrs, sizes, ctens, coff = [], [], [], []
        for i, p in enumerate(plist):
            st = self.state[p]
            ptrs += [p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()]
            n = p.numel()
            sizes.append(n)
            for off in range(0, n, chunk):
                ctens.append(i)
                coff.append(off)
        tab = (torch.tensor(ptrs, dtype=torch.int64, device=dev), torch.tensor(sizes, dtype=torch.int64, device=dev),
               torch.tensor(ctens, dtype=torch.int32, device=dev), torch.tensor(coff, dtype=torch.int64, device=dev),
               len(plist), len(ctens))
        self._tables[gi] = (key, tab)
        return tab

    def _hyper(self, gi, group, dev):
        """Device-side {step, lr}: the kernel advances step itself; lr is refreshed only when the host value changed
        (a fill outside any captured graph), so LR decay needs no re-capture."""
        h = self._hypers.get(gi)
        if h is None:
            t = torch.tensor([float(group["step"] - 1), float(group["lr"])], dtype=torch.float32, device=dev)
            h = self._hypers[gi] = [t, group["lr"]]
        elif h[1] != group["lr"]:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("learning rate changed during graph capture")
            h[0][1:].fill_(float(group["lr"]))
            h[1] = group["lr"]
        return h[0]

    # ---------------------------------------------------------------- deferred per-layer update
    def attach_layer(self, layer, weight):
        """Take `weight` (the master of cpcsv.modules.KernelLayer `layer`) out of the multi-tensor kernel: its backward
        passes only ACCUMULATE into the layer's fp32 accumulator (no per-call unpack), and step() runs ONE fused launch
        for it - accumulator -> gradient (+ spectral-norm terms) -> Adam -> operand copies (cpcsv_layer_update).
        Replaces unpack x calls + Adam + pack_fwd + pack_bwd (~56 B of HBM traffic per parameter) by ~36 B."""
        if id(weight) in self._fused_ids:
            return
        st = self.state[weight]
        if not st:
            st["exp_avg"] = torch.zeros_like(weight)
            st["exp_avg_sq"] = torch.zeros_like(weight)
        self._fused_ids.add(id(weight))
        ent = [layer, weight, None]
        self._layers.append(ent)
        self._ent_of[id(layer)] = ent
        layer.fused, layer.fused_terms, layer.fused_opt = True, [], self
        self._tables.clear()

    # -- in-backward form: the launch of a layer goes out from its last weight-gradient call of the step ------------
    def prepare_step(self):
        """Call where the step's gradients start to accumulate (right after the bucket was zeroed). Publishes the step
        count the coming optimiser step will use, so that layer updates launched from inside the backward pass apply the
        same bias corrections as the multi-tensor launch that follows in step()."""
        h = self._hypers.get(0)
        if not self.inline or h is None or not self._layers:
            return
        # the in-backward launches read the optimiser's own {step, lr} scalars with step_add = 1 (cpcsv_update_desc): no copy
        self._hyper_next = h[0]
        self._prepared = True

    def inline_ok(self):
        return self.inline and self._prepared

    def update_layer_now(self, layer):
        from . import modules as M
        ent = self._ent_of[id(layer)]
        K.layer_update(self._update_desc(ent, self.param_groups[0], self._hyper_next, step_add=1.0))
        layer.fused_keep = list(layer.fused_terms)
        layer.fused_terms.clear()
        layer.fused_seen, layer.fused_updated = 0, True
        if M.UPDATE_LOG is not None:
            M.UPDATE_LOG.append(layer)

    def flush_stashes(self):
        """Run the parked half of every shared weight-gradient launch whose partner never came: call BEFORE the gradient
        exchange, so that the exchanged accumulators are complete."""
        from .functional import flush_stash
        for ent in self._layers:
            flush_stash(ent[0])

    def is_fused(self, p):
        return id(p) in self._fused_ids

    def _update_desc(self, ent, group, hyper, gscale=1.0, step_add=0.0, wire_of=None):
        layer, weight, d = ent
        dt = layer.fused_dt
        fwd, bwd, lin = layer._pack_bufs[(dt, weight.device)]
        if d is None:
            from . import functional as F
            from .runtime import pad8
            d = ent[2] = L.UpdateDesc()
            st = self.state[weight]
            d.p, d.m, d.v = weight.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            d.dtype, d.Cout, d.Cin, d.taps, d.S = dt, layer.cout, layer.cin, layer.taps, layer.slices
            d.Cin_s, d.Cout_s, d.sum = layer.cin_s, pad8(layer.cout), int(layer.subpixel)
            for i in range(L.MAX_TAPS):
                d.tapmap[i] = i if i < layer.slices else -1
                d.masks[i] = F.SUB_MASKS[i] if (layer.subpixel and i < 16) else 0
            b1, b2 = group["betas"]
            d.beta1, d.beta2, d.eps = b1, b2, group["eps"]
        d.G, d.g_bf16 = layer._g.data_ptr(), 0
        if wire_of is not None:
            # data-parallel exchange with the bf16 payload: the SUM-reduced values live in the bucket's bf16 wire buffer, at the
            # accumulator's element offsets (cpcsv.dist.GradBucket.reduce_extra_async)
            bi, lo, _ = layer._g_span
            w = wire_of(bi)
            if w is not None:
                d.G, d.g_bf16 = w.data_ptr() + 2 * lo, 1
        d.fwd, d.bwd, d.lin = fwd.data_ptr(), (bwd.data_ptr() if bwd is not None else None), (lin.data_ptr() if lin is not None else None)
        d.hyper = hyper.data_ptr()
        d.gscale = gscale
        d.step_add = step_add
        terms = layer.fused_terms
        if len(terms) > 4:
            raise RuntimeError("%s: %d spectral-norm calls in one step (at most 4 supported)" % (layer.name, len(terms)))
        d.nterms = len(terms)
        for k, (gw, sigma, u, v) in enumerate(terms):
            d.gw[k], d.sigma[k], d.u[k], d.v_sn[k] = gw.data_ptr(), sigma.data_ptr(), u.data_ptr(), v.data_ptr()
        return d

    def _step_layers(self, group, hyper, pending=None, gscale=1.0, wire_of=None):
        # One stream by default. Fanning the (independent) layers out over side streams was measured and dropped: the
        # launches are HBM-bound together (G: 3.2 GB in 1.47 ms either way) and every extra stream cost the critic phase
        # ~2 ms of cross-stream waits (CPCSV_UPDATE_STREAMS=n re-enables it for experiments).
        import os
        nstreams = int(os.environ.get("CPCSV_UPDATE_STREAMS", "1"))
        capturing = torch.cuda.is_current_stream_capturing()
        side = []
        if nstreams > 1 and len(self._layers) > 1 and not capturing and not pending:      # (the chunk waits below are on THIS stream)
            if not hasattr(self, "_side"):
                self._side = [torch.cuda.Stream() for _ in range(nstreams - 1)]
                self._order = sorted(range(len(self._layers)), key=lambda i: -self._layers[i][1].numel())
            side = self._side
            cur = torch.cuda.current_stream()
            for st in side:
                st.wait_stream(cur)
        order = getattr(self, "_order", range(len(self._layers))) if side else range(len(self._layers))
        from .functional import flush_stash
        waited = 0
        for n, idx in enumerate(order):
            ent = self._layers[idx]
            if ent[0].fused_updated:                 # already applied from inside this step's backward pass
                continue
            flush_stash(ent[0])
            if pending:
                # chunks of the accumulator exchange still on the wire (dist.GradBucket.reduce_extra_async): this layer's
                # update may start once every chunk up to the end of ITS accumulator has landed
                # pending = [(extra buffer index, lo, hi, wait)] in issue order = (buffer, offset) order
                span = getattr(ent[0], "_g_span", None)
                if span is None:
                    raise RuntimeError("%s: deferred-update layer without its accumulator span (GANTrainer._attach_deferred_updates)" % ent[0].name)
                bi, _, end = span
                while waited < len(pending) and (pending[waited][0], pending[waited][1]) < (bi, end):
                    pending[waited][3]()
                    waited += 1
            d = self._update_desc(ent, group, hyper, gscale, wire_of=wire_of if pending else None)
            k = n % (len(side) + 1)
            if side and k:
                with torch.cuda.stream(side[k - 1]):
                    K.layer_update(d)
            else:
                K.layer_update(d)
        if pending:
            while waited < len(pending):
                pending[waited][3]()
                waited += 1
        if side:
            for st in side:
                torch.cuda.current_stream().wait_stream(st)
        for ent in self._layers:
            layer, weight, _ = ent
            if not layer.fused_updated:
                layer.fused_keep = list(layer.fused_terms)      # tensors the launch reads stay alive until the next step
                if not layer.fused_expected and layer.fused_seen:
                    layer.fused_expected = layer.fused_seen      # weight-gradient calls per step, learnt in the first step
            layer.fused_terms.clear()
            layer.fused_seen, layer.fused_updated = 0, False
            weight._cpcsv_epoch = getattr(weight, "_cpcsv_epoch", 0) + 1
            layer.mark_packed(weight, layer.fused_dt, ("fwd", "bwd"))
        self._prepared = False

    def export_grad(self, p, wire_of=None):
        """Gradient of `p` in master layout - for tests and diagnostics. Deferred-update weights have no materialised
        .grad: it is rebuilt here from the accumulator (the standalone unpack kernels, accumulator left untouched).
        wire_of: after a data-parallel exchange with the bf16 payload the REDUCED values live in the bucket's wire buffer
        (cpcsv.dist.GradBucket.wire_of), not in the fp32 accumulator."""
        if id(p) not in self._fused_ids:
            return p.grad
        from . import functional as F
        layer = next(l for l, w, _ in self._layers if w is p)
        g = layer._g
        if wire_of is not None and getattr(layer, "_g_span", None) is not None:
            bi, lo, hi = layer._g_span
            w = wire_of(bi)
            if w is not None:
                g = w[lo:hi].float().view_as(layer._g)
        out = torch.zeros_like(p)
        if layer.subpixel:
            K.unpack_wgrad_sum(g, out, layer.cout, layer.cin, 9, 16, F.SUB_MASKS, layer.cin_s, False, rezero=0)
        else:
            K.unpack_wgrad(g, out, None, None, None, None, layer.cout, layer.cin, layer.taps, layer.slices, layer.tapmap,
                           layer.cin_s, False, rezero=0)
        for gw, sigma, u, v in (layer.fused_terms or (getattr(layer, "fused_keep", []) if layer.fused_updated else [])):
            out -= (gw[0] / (sigma[0] * sigma[0])) * torch.outer(u, v).view_as(out)
        return out

    def sync_lr(self):
        """Push host-side lr changes to the device scalars (call after editing param_groups when using HIP graphs)."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._hypers and self._hypers[gi][1] != group["lr"]:
                self._hypers[gi][0][1:].fill_(float(group["lr"]))
                self._hypers[gi][1] = group["lr"]

    @torch.no_grad()
    def step(self, closure=None, pending=None, gscale=1.0, wire_of=None):
        """pending / gscale: data-parallel runs - the chunks of the accumulator exchange that are still in flight and the 1/world
        the SUM-reduced accumulators still need (see _step_layers); wire_of(buffer index) -> the bf16 wire buffer holding the reduced
        values instead of the fp32 accumulator (bf16 payload), or None."""
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None and id(p) not in self._fused_ids]
            if not plist:
                if self._layers:
                    raise RuntimeError("FusedAdam needs at least one multi-tensor parameter per group (it advances the step count)")
                continue
            for p in plist:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous fp32 GPU parameters and gradients")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            group["step"] = group.get("step", 0) + 1          # host mirror (not advanced by graph replays)
            hyper = self._hyper(gi, group, plist[0].device)
            ptrs, sizes, ctens, coff, nt, nchunks = self._table(gi, plist)
            b1, b2 = group["betas"]
            K.adam_step(ptrs, sizes, nt, nchunks, ctens, coff, hyper, b1, b2, group["eps"])
            for p in plist:      # invalidate packed-operand caches (cpcsv.modules.KernelLayer.packs)
                p._cpcsv_epoch = getattr(p, "_cpcsv_epoch", 0) + 1
            if gi == 0 and self._layers:
                self._step_layers(group, hyper, pending, gscale, wire_of)        # reads the step count the launch above just advanced
        return loss
