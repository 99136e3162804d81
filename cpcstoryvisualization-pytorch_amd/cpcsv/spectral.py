"""Spectral-norm power iterations of MANY layers per launch (cpcsv_spectral_sigma_multi).

torch.nn.utils.spectral_norm (old hook API; reference model.py:19,79,502-510,544-552,583-594) runs one power iteration per
forward call of every wrapped layer: 54 evaluations per training step, three launches each, all of them depending only on
the weights. The trainer knows the call pattern of a step (D step tower(real), tower(fake), head(real),
head(wrong), head(fake); G step tower(fake), head(fake)), so a `SpectralPlan` (one per critic) evaluates them up front in
rounds: round k runs iteration k of every layer that is called more than k times, ONE launch triple per round. The layers then consume the
precomputed (sigma, u, v) of their k-th call in call order (cpcsv.modules.Conv2d.spectral_state), which keeps the
reference's per-layer u/v sequence exactly.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import kernels as K
from . import runtime

SLOTS = 4          # calls of one layer whose (sigma, u, v) can be live at once: <= 3 in the D step + 1 in the G step
# ONE pass over every W per iteration (cpcsv_spectral_sigma_multi1: column slabs held in registers, fixed-order partial sums) instead
# of two; CPCSV_SN_ONEPASS=0: the two-pass form (A/B). The reproducible mode always takes the two-pass form (its summation orders).
_ONEPASS = os.environ.get("CPCSV_SN_ONEPASS", "1") != "0"


class SpectralPlan:
    def __init__(self, layers):
        """layers: [(holder, calls in the D step, calls in the G step)]; holder = cpcsv.modules.Conv2d / Linear with
        spectral=True, already on the GPU."""
        self.layers = [(h, d, g) for h, d, g in layers if getattr(h, "spectral", False)]
        self._tables = {}
        for h, d, g in self.layers:
            if d + g > SLOTS:
                raise RuntimeError("SpectralPlan: %d calls of one layer per step (at most %d)" % (d + g, SLOTS))
            rows, cols = h._sn_shape
            w = h.master()
            h._sn_slots = torch.zeros(SLOTS, 2 + rows + cols, dtype=torch.float32, device=w.device)
            if h._sn_work is None or h._sn_work.device != w.device:
                h._sn_work = torch.zeros(rows + cols + 2, dtype=torch.float32, device=w.device)
            h._sn_queue = []

    def _round(self, phase, r):
        """(jobs tensor, njobs, start1, nblk1, start2, nblk2, holders) of round r of `phase`, cached per reduction mode."""
        key = (phase, r, runtime.deterministic())
        tab = self._tables.get(key)
        if tab is not None:
            return tab
        members = [(h, (r if phase == "D" else d + r)) for h, d, g in self.layers if (d if phase == "D" else g) > r]
        if not members:
            self._tables[key] = None
            return None
        dev = members[0][0].master().device
        jobs = (L.SnJob * len(members))()
        s1, s2 = [0], [0]
        for i, (h, slot) in enumerate(members):
            rows, cols = h._sn_shape
            jobs[i].w, jobs[i].u, jobs[i].v = h.master().data_ptr(), h.weight_u.data_ptr(), h.weight_v.data_ptr()
            jobs[i].work, jobs[i].out = h._sn_work.data_ptr(), h._sn_slots[slot].data_ptr()
            jobs[i].rows, jobs[i].cols = rows, cols
            s1.append(s1[-1] + K.sn_multi_blocks(rows, cols, 1))
            s2.append(s2[-1] + K.sn_multi_blocks(rows, cols, 2))
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        one = None
        max_rows = max(h._sn_shape[0] for h, _ in members)
        if _ONEPASS and not runtime.deterministic() and max_rows <= 1024:
            s3, off = [0], [0]
            for h, _ in members:
                rows, cols = h._sn_shape
                nb = K.sn_multi_blocks(rows, cols, 3)
                s3.append(s3[-1] + nb)
                off.append(off[-1] + nb * rows + nb)        # the slabs' y shares [nb][rows], then their |t|^2 shares [nb]
            one = (torch.tensor(s3, dtype=torch.int32, device=dev), s3[-1], torch.empty(off[-1], dtype=torch.float32, device=dev),
                   torch.tensor(off[:-1], dtype=torch.int64, device=dev), max_rows)
        tab = (raw, len(members), torch.tensor(s1, dtype=torch.int32, device=dev), s1[-1],
               torch.tensor(s2, dtype=torch.int32, device=dev), s2[-1], members,
               tuple((h.master().data_ptr(), h.weight_u.data_ptr(), h.weight_v.data_ptr()) for h, _ in members), one)
        self._tables[key] = tab
        return tab

    def run(self, phase):
        """Enqueue every power iteration `phase` ("D" | "G") will consume, on the current stream; arm the layers' queues."""
        for h, d, g in self.layers:
            h._sn_queue = []
        r = 0
        while True:
            tab = self._round(phase, r)
            if tab is None:
                break
            raw, n, s1, n1, s2, n2, members, ptrs, one = tab
            if ptrs != tuple((h.master().data_ptr(), h.weight_u.data_ptr(), h.weight_v.data_ptr()) for h, _ in members):
                self._tables.clear()                      # a parameter or buffer was re-allocated (load_state_dict(assign=True), .to())
                return self.run(phase)
            with torch.no_grad():
                if one is not None:
                    K.spectral_sigma_multi1(raw, n, one[0], one[1], one[2], one[3], one[4])
                else:
                    K.spectral_sigma_multi(raw, n, s1, n1, s2, n2, True)
            for h, slot in members:
                rows, cols = h._sn_shape
                o = h._sn_slots[slot]
                h._sn_queue.append((o[:2], o[2:2 + rows], o[2 + rows:]))
            r += 1

    def disarm(self):
        for h, d, g in self.layers:
            h._sn_queue = []


def plan_for_critic(net):
    """SURVEY A12 call pattern of a critic: tower layers (`encode_img.*`) 2 calls in the D step and 1 in the G step, head
    layers (`get_cond_logits.*`) 3 and 1. The optional order critic (`seq_consisten_model.*`) is not planned: its layers
    evaluate on the fly. One plan per critic: its rounds run on that critic's own stream, off the generator's critical path."""
    layers = []
    if net is not None:
        for name, m in net.named_modules():
            if not getattr(m, "spectral", False) or not hasattr(m, "_sn_shape"):
                continue
            if name.startswith("encode_img."):
                layers.append((m, 2, 1))
            elif name.startswith("get_cond_logits."):
                layers.append((m, 3, 1))
    return SpectralPlan(layers) if layers else None
