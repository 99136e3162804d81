"""F4 input pipeline: the device-side normalisation of pre-decoded uint8 frames against the restated
loader transform chain of the reference (main_pororo.py:71-92; oracle/cpcsv_oracle/ingest.py), bit for bit."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu


def test_oracle_transform_matches_fixture_and_closed_form():
    from oracle.cpcsv_oracle.ingest import image_transform, video_transform
    fx = gu.load("ingest.npz")
    got = np.stack([video_transform(v).numpy() for v in fx["story/u8"]])
    assert got.shape == (2, 3, 3, 16, 16) and np.array_equal(got, fx["story/out"])
    assert np.array_equal(np.stack([image_transform(f).numpy() for f in fx["seg/u8"]]), fx["seg/out"])
    # (x/255 - .5)/.5 spans [-1, 1] and is monotone in the byte value
    lut = image_transform(np.arange(256, dtype=np.uint8).reshape(16, 16)).flatten()
    assert lut[0] == -1.0 and lut[255] == 1.0 and bool((lut[1:] > lut[:-1]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["story", "image", "seg", "image_stats"])
def test_device_ingest_is_bit_exact(what):
    from cpcsv import ingest
    fx = gu.load("ingest.npz")
    kw = {}
    key = what
    if what == "image_stats":
        key, kw = "image", dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
    u8 = torch.from_numpy(fx[key + "/u8"]).cuda()
    want = torch.from_numpy(fx["image/out_stats" if what == "image_stats" else key + "/out"])
    for dt in (torch.float32, torch.bfloat16):
        out, nhwc = ingest.normalise_u8(u8, want_nhwc=True, nhwc_dtype=dt, **kw)
        assert out.dtype == torch.float32 and tuple(out.shape) == tuple(want.shape)
        assert torch.equal(out.cpu(), want), (what, (out.cpu() - want).abs().max())
        # NHWC frames: same values (rounded once for bf16), zero channel pads
        frames = want.permute(0, 2, 1, 3, 4).reshape(-1, *want.shape[1:2], *want.shape[3:]) if want.dim() == 5 else want
        c = frames.shape[1]
        ref = frames.permute(0, 2, 3, 1).to(dt)
        assert torch.equal(nhwc[..., :c].cpu(), ref) and float(nhwc[..., c:].abs().max()) == 0.0


@pytest.mark.gpu
def test_to_device_batch_accepts_the_reference_datasets_uint8_frames():
    """The reference's StoryDataset already returns `images_numpy` (T,H,W,C uint8, datasets/pororo.py:139,150): a
    pre-decoded loader can drop the CPU transform and ship only that."""
    from cpcsv import ingest
    fx = gu.load("ingest.npz")
    batch = {"images_numpy": torch.from_numpy(fx["story/u8"]), "labels": torch.ones(2, 3, 9), "text": ["a", "b"]}
    dev = ingest.to_device_batch(batch, "cuda")
    assert torch.equal(dev["images"].cpu(), torch.from_numpy(fx["story/out"])) and dev["text"] == ["a", "b"] and dev["labels"].is_cuda
    batch = {"images": torch.from_numpy(fx["image/u8"]), "images_seg": torch.from_numpy(fx["seg/u8"])}
    dev = ingest.to_device_batch(batch, "cuda")
    assert torch.equal(dev["images"].cpu(), torch.from_numpy(fx["image/out"]))
    assert torch.equal(dev["images_seg"].cpu(), torch.from_numpy(fx["seg/out"]))
