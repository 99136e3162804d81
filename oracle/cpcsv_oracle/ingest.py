"""Input transform chain of the reference's loaders, restated. TEST INFRASTRUCTURE ONLY.

Restates /root/reference/main_pororo.py:71-92: per frame `PIL.Image.fromarray -> transforms.Resize((IMSIZE, IMSIZE)) ->
transforms.ToTensor() -> transforms.Normalize((.5,.5,.5), (.5,.5,.5))`, and `video_transform` (stack the T frames,
permute to (C,T,H,W)). torchvision 0.4.2 (requirements.txt:46) is NOT installed in this image and not vendored in
/root/reference, so its two arithmetic ops are restated from their published definitions:
    ToTensor  (uint8 HWC ndarray): torch.from_numpy(pic.transpose(2,0,1)).float().div(255)
    Normalize                    : tensor.sub_(mean[:,None,None]).div_(std[:,None,None])
PARITY UNPINNED for these two torchvision ops (no reference-side run is possible here); frames are taken at the training
resolution, where Resize is the identity.
"""
import numpy as np
import torch


def to_tensor(pic_u8):
    """transforms.ToTensor on an (H,W,C) or (H,W) uint8 array."""
    pic = np.asarray(pic_u8)
    if pic.ndim == 2:
        pic = pic[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(pic.transpose(2, 0, 1))).float().div(255)


def normalize(t, mean, std):
    """transforms.Normalize (in place on a clone, like the reference chain's fresh tensor)."""
    t = t.clone()
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    return t.sub_(m).div_(s)


def image_transform(frame_u8, mean=None, std=None):
    """main_pororo.py:71-84 on one frame already at IMSIZE x IMSIZE."""
    t = to_tensor(frame_u8)
    c = t.shape[0]
    return normalize(t, mean or (0.5,) * c, std or (0.5,) * c)


def video_transform(video_u8, mean=None, std=None):
    """main_pororo.py:86-92: T x H x W x C uint8 -> (C, T, H, W) fp32."""
    return torch.stack([image_transform(f, mean, std) for f in video_u8]).permute(1, 0, 2, 3)
