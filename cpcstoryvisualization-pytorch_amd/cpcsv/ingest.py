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


def to_device_batch(batch, device, non_blocking=True):
    """One loader batch (dict) -> device tensors the step reads. uint8 image tensors (`images`, `images_seg` given as
    HWC uint8, or the reference dataset's own `images_numpy`) are normalised on the device; everything else is copied
    as is; the `text` strings stay on the host."""
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
