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


@pytest.mark.gpu
def test_device_feeder_rotating_pinned_slots_deliver_every_batch_intact():
    """cpcsv.ingest.DeviceFeeder (the host half of F4: pinned staging, own copy stream, one batch of look-ahead in
    GANTrainer.train): ten different batches through three rotating slot sets, each consumed only after wait_ready() while the
    NEXT one is already being staged - every tensor arrives bit-exact (a slot rewritten before its copy had read it, or a batch
    read before its copy landed, would show as a mix of two batches), uint8 frames arrive normalised, strings stay on the host."""
    from cpcsv import ingest
    fx = gu.load("ingest.npz")
    feeder = ingest.DeviceFeeder("cuda", slots=3)
    g = torch.Generator().manual_seed(5)
    batches = []
    for i in range(10):
        batches.append({"images": torch.rand(12, 3, 5, 64, 64, generator=g), "description": torch.randn(12, 5, 356, generator=g),
                        "labels": (torch.rand(12, 5, 9, generator=g) < 0.3).float(), "text": ["story %d" % i] * 12})
    burn = torch.randn(4096, 4096, device="cuda")
    pending = feeder.put(batches[0])
    for i in range(10):
        cur = pending
        pending = feeder.put(batches[i + 1]) if i + 1 < 10 else None      # look-ahead: staged while `cur` is consumed
        ingest.wait_ready(cur)
        assert "_ready" not in cur and cur["text"] == batches[i]["text"]
        burn = burn @ burn * 1e-4                                          # keep the consuming stream busy behind the wait
        for k in ("images", "description", "labels"):
            assert torch.equal(cur[k].cpu(), batches[i][k]), (i, k)
    u8 = feeder.put({"images": torch.from_numpy(fx["image/u8"]), "images_seg": torch.from_numpy(fx["seg/u8"])})
    ingest.wait_ready(u8)
    assert torch.equal(u8["images"].cpu(), torch.from_numpy(fx["image/out"]))
    assert torch.equal(u8["images_seg"].cpu(), torch.from_numpy(fx["seg/out"]))
    dev = ingest.to_device_batch({"labels": torch.ones(2, 3)}, "cuda", feeder=True)
    assert "_ready" in dev and ingest.wait_ready(dev)["labels"].is_cuda


@pytest.mark.gpu
def test_device_feeder_keeps_its_pinned_buffers_across_alternating_batch_kinds():
    """GANTrainer.train() feeds ONE feeder two alternating batch kinds - the image batch and the story batch of a step (reference
    trainer.py:250-252) - whose tensors share key names but not shapes. Every (key, shape, dtype) keeps its own page-locked
    buffer per slot: once each kind has visited each slot nothing is allocated any more, and every batch still arrives intact."""
    from cpcsv import ingest
    feeder = ingest.DeviceFeeder("cuda", slots=3)
    g = torch.Generator().manual_seed(9)
    mk_im = lambda: {"images": torch.rand(6, 3, 64, 64, generator=g), "description": torch.randn(6, 356, generator=g)}
    mk_st = lambda: {"images": torch.rand(2, 3, 5, 64, 64, generator=g), "description": torch.randn(2, 5, 356, generator=g)}
    seen = []
    for i in range(12):
        host = mk_im() if i % 2 == 0 else mk_st()
        dev = ingest.wait_ready(feeder.put(host))
        for k in host:
            assert torch.equal(dev[k].cpu(), host[k]), (i, k)
        seen.append(feeder.allocs)
    # 2 kinds x 2 keys x 3 slots = 12 buffers at most, all of them created within the first 6 puts (each kind through each slot)
    assert seen[5] <= 12 and seen[-1] == seen[5], seen
