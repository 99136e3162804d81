"""F4 input pipeline, device half: pre-decoded uint8 frames -> the batch tensors the step reads.

The reference decodes PNGs with PIL and runs `transforms.ToTensor` + `transforms.Normalize(.5, .5)` per frame on CPU
worker processes (/root/reference/main_pororo.py:71-92, datasets/pororo.py:103-151); at >= 3 k story-frames/s per GPU that
loader cannot feed 8 GPUs. Here the loader hands over the uint8 frames it already has (`images_numpy`,
datasets/pororo.py:139: T x H x W x C) - a quarter of the fp32 bytes over PCIe - and ONE kernel (cpcsv_ingest_u8) produces
the normalised fp32 channel-planar batch tensor (bit-equal to the reference's transform chain) and, on request, the NHWC
compute-dtype frames. Resizing is not done here: frames are expected at the training resolution (cfg.IMSIZE / cfg.SESIZE).
"""
import torch

from . import kernels as K
from .runtime import pad8, require_gpu, tdtype

_CONST = {}


def _stats(dev, c, mean, std):
    key = (dev, c, tuple(mean), tuple(std))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = (torch.tensor(list(mean), dtype=torch.float32, device=dev),
                           torch.tensor(list(std), dtype=torch.float32, device=dev))
    return t


def normalise_u8(frames, mean=None, std=None, want_nhwc=False, nhwc_dtype=None):
    """frames: uint8 device tensor, (B,T,H,W,C) for stories -> fp32 (B,C,T,H,W) (`video_transform`'s layout),
    or (B,H,W,C) / (B,H,W) for single images / segmentation maps -> fp32 (B,C,H,W).
    Returns the fp32 tensor, or (fp32, nhwc frames [B*T,H,W,Cs]) when want_nhwc."""
    require_gpu(frames)
    if frames.dtype != torch.uint8:
        raise TypeError("normalise_u8 expects uint8 frames, got %s" % frames.dtype)
    if frames.dim() == 3:
        frames = frames.unsqueeze(-1)
    frames = frames.contiguous()
    story = frames.dim() == 5
    if story:
        b, t, h, w, c = frames.shape
        out = torch.empty((b, c, t, h, w), dtype=torch.float32, device=frames.device)
        sb, st, sc = c * t * h * w, h * w, t * h * w
    else:
        b, h, w, c = frames.shape
        t = 1
        out = torch.empty((b, c, h, w), dtype=torch.float32, device=frames.device)
        sb, st, sc = c * h * w, 0, h * w
    mean = mean if mean is not None else (0.5,) * c
    std = std if std is not None else (0.5,) * c
    m, s = _stats(frames.device, c, mean, std)
    nh = None
    if want_nhwc:
        nh = torch.empty((b * t, h, w, pad8(c)), dtype=nhwc_dtype or tdtype(), device=frames.device)
    K.ingest_u8(frames, out, nh, b * t, t, sb, st, sc, c, h * w, pad8(c), m, s)
    return (out, nh) if want_nhwc else out


class DeviceFeeder:
    """Pinned, rotating host->device staging for loader batches (the host half of F4: the reference's `.cuda()` on pageable
    loader tensors, trainer.py:254-274, is a synchronous staged copy per tensor). A batch is memcpy'd into one of `slots`
    sets of page-locked buffers (allocated once per key / shape / dtype) and moved with asynchronous copies on a dedicated copy stream,
    so the transfer of batch N+1 - GANTrainer.train() fetches one batch ahead - overlaps step N; uint8 frames are normalised
    on the device right behind their copy (a quarter of the fp32 bytes over PCIe). A slot is rewritten only after the event
    recorded behind its last copy has completed. The consumer calls `wait_ready(batch)` on the stream that will read it."""

    def __init__(self, device, slots=3):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.slots = [{} for _ in range(max(2, slots))]
        self.events = [None] * len(self.slots)
        self.k = 0
        self.bytes = 0
        self.allocs = 0                      # pinned buffers created so far (tests: stays flat once every batch kind has been seen)

    def put(self, batch):
        slot, ev = self.slots[self.k], self.events[self.k]
        if ev is not None:
            ev.synchronize()                         # its copies are `slots` batches old: normally long done
        out = {}
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if k == "text" or not torch.is_tensor(v):
                    out[k] = v
                    continue
                if v.is_cuda:
                    out[k] = v
                    continue
                # one page-locked buffer per (key, shape, dtype) and slot: GANTrainer.train() feeds two alternating batch kinds
                # (image batch, story batch) whose tensors share key names but not shapes - keyed by name alone every put would
                # re-allocate (pageable alloc + copy + pinning, ~6 MB per step on the loader thread)
                pk = (k, tuple(v.shape), v.dtype)
                pin = slot.get(pk)
                if pin is None:
                    pin = slot[pk] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                    self.allocs += 1
                pin.copy_(v)
                self.bytes += pin.numel() * pin.element_size()
                d = pin.to(self.device, non_blocking=True)
                if k in ("images", "images_seg") and d.dtype == torch.uint8:
                    d = normalise_u8(d)
                out[k] = d
            if "images" not in out and torch.is_tensor(out.get("images_numpy")) and out["images_numpy"].dtype == torch.uint8:
                out["images"] = normalise_u8(out["images_numpy"])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.events[self.k] = ev
        self.k = (self.k + 1) % len(self.slots)
        out["_ready"] = ev
        return out


def wait_ready(batch):
    """Make the current stream wait for a DeviceFeeder batch's copies (and tell the allocator that this stream reads them)."""
    ev = batch.pop("_ready", None) if isinstance(batch, dict) else None
    if ev is not None:
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
    return batch


_FEEDERS = {}


def to_device_batch(batch, device, non_blocking=True, feeder=False):
    """One loader batch (dict) -> device tensors the step reads. uint8 image tensors (`images`, `images_seg` given as
    HWC uint8, or the reference dataset's own `images_numpy`) are normalised on the device; everything else is copied
    as is; the `text` strings stay on the host. With `feeder=True` (what GANTrainer.train() asks for) the copies go through a
    DeviceFeeder (pinned staging, own copy stream): the returned dict then carries a `_ready` event and must be passed through
    wait_ready() on the consuming stream before use."""
    dev = torch.device(device)
    if feeder and dev.type == "cuda":
        f = _FEEDERS.get(dev)
        if f is None:
            f = _FEEDERS[dev] = DeviceFeeder(dev)
        return f.put(batch)
    out = {}
    for k, v in batch.items():
        if k == "text" or not torch.is_tensor(v):
            out[k] = v
            continue
        v = v.to(device, non_blocking=non_blocking)
        if k in ("images", "images_seg") and v.dtype == torch.uint8:
            v = normalise_u8(v)
        out[k] = v
    if "images" not in out and "images_numpy" in out and torch.is_tensor(out["images_numpy"]) and out["images_numpy"].dtype == torch.uint8:
        out["images"] = normalise_u8(out["images_numpy"])
    return out
