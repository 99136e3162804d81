#!/usr/bin/env python3
"""tests/golden/ingest.npz: uint8 frames and what the reference's loader transform chain makes of them
(oracle/cpcsv_oracle/ingest.py restates main_pororo.py:71-92; torchvision itself is not installed here, see that header:
PARITY UNPINNED for ToTensor / Normalize). Inputs cover every byte value; sizes are small on purpose (fixture data only)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.cpcsv_oracle.ingest import image_transform, video_transform  # noqa: E402


def main():
    rng = np.random.RandomState(7)
    story = rng.randint(0, 256, size=(2, 3, 16, 16, 3)).astype(np.uint8)        # (B, T, H, W, C): `images_numpy`, pororo.py:139
    story.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)                    # every byte value occurs
    image = rng.randint(0, 256, size=(3, 16, 16, 3)).astype(np.uint8)
    seg = rng.randint(0, 256, size=(3, 16, 16)).astype(np.uint8)
    seg.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)[::-1]
    fx = {"story/u8": story, "image/u8": image, "seg/u8": seg,
          "story/out": np.stack([video_transform(v).numpy() for v in story]),    # (B, C, T, H, W)
          "image/out": np.stack([image_transform(f).numpy() for f in image]),    # (B, 3, H, W)
          "seg/out": np.stack([image_transform(f).numpy() for f in seg]),        # (B, 1, H, W)
          # a non-default normalisation, to catch a kernel that hard-codes 0.5
          "image/out_stats": np.stack([image_transform(f, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)).numpy() for f in image])}
    path = os.path.join(REPO, "tests", "golden", "ingest.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
