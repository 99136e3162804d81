"""Multi-tensor Adam on the HIP path (torch.optim.Adam semantics, reference trainer.py:212-220).

One kernel launch updates every parameter of an optimiser: a device-side pointer table
{p, g, m, v} per tensor plus a chunk map (block -> tensor, offset). torch's per-tensor foreach
path would be hundreds of launches for G's ~110 tensors; the reference's Adam is 1.8 % of its CPU step
and 4.4 GB/step of pure HBM streaming on the GPU, so it is one streaming pass here.
"""
import torch

from . import kernels as K


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}
        self._hypers = {}

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

    def sync_lr(self):
        """Push host-side lr changes to the device scalars (call after editing param_groups when using HIP graphs)."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._hypers and self._hypers[gi][1] != group["lr"]:
                self._hypers[gi][0][1:].fill_(float(group["lr"]))
                self._hypers[gi][1] = group["lr"]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
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
        return loss
