"""The drop-in entry point itself: GANTrainer(output_dir, args).train(imageloader, storyloader, testloader, stage)
(reference trainer.py:187-485, called from main_pororo.py:137-138) on fake loaders, then the checkpoint wire format
(reference miscc/utils.py:323-338; resume trainer.py:121-131) and the eval-mode forward (reference inference.py:77-81,
143-199: BN running statistics, frozen spectral norm) against the oracle."""
import os
import types

import pytest
import torch

from tests import golden_util as gu
from tests import parity_util as pu

pytestmark = pytest.mark.gpu


class FakeLoader:
    """A DataLoader stand-in: len() and iteration over CPU batch dicts (the reference's loaders yield dicts of CPU
    tensors plus a 'text' list, trainer.py:254-274)."""

    def __init__(self, batches):
        self.batches = batches

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def _loaders(oc, n_story=3, n_image=2):
    from oracle.cpcsv_oracle import synthetic_batch
    st, im = [], []
    for i in range(n_story):
        s, _ = synthetic_batch(oc, seed=100 + i)
        s["text"] = ["story %d" % i] * oc.st_batch
        st.append(s)
    for i in range(n_image):                       # fewer image batches than steps: sample_real_image_batch wraps around
        _, m = synthetic_batch(oc, seed=200 + i)
        m["text"] = ["image %d" % i] * oc.im_batch
        im.append(m)
    return FakeLoader(im), FakeLoader(st)


def _setup_cfg(oc, max_epoch=3):
    cfg = pu.apply_cfg(oc)
    cfg.TRAIN.FLAG = True
    cfg.TRAIN.MAX_EPOCH = max_epoch
    cfg.TRAIN.SNAPSHOT_INTERVAL = 1
    cfg.TRAIN.LR_DECAY_EPOCH = 1
    cfg.NET_G = ''
    return cfg


def test_train_loop_checkpoints_resume_and_eval_mode(tmp_path):
    from cpcsv import runtime
    import trainer as T
    runtime.set_compute_dtype("fp32")
    was = runtime.set_deterministic(True)
    try:
        fx = gu.load("step_plain.npz")
        oc = gu.cfg_of(fx)
        cfg = _setup_cfg(oc)
        args = types.SimpleNamespace(cfg_file=None, continue_ckpt=None)
        out_dir = str(tmp_path / "run")

        # ---- (1) train(): 3 epochs x 3 story batches through the real loop (default launch mode: piecewise graphs).
        # LR_DECAY_EPOCH=1: halved after epoch 1, the interval doubles, halved again after epoch 2 (reference :447-456)
        tr = T.GANTrainer(out_dir, args, ratio=1.0)
        imageloader, storyloader = _loaders(oc)
        torch.manual_seed(11)                      # weights (built inside train()) and the device noise stream
        tr.train(imageloader, storyloader, None, 1)
        torch.cuda.synchronize()
        init = {k: [p.detach().clone() for p in n.parameters()] for k, n in zip("G im st se".split(), tr.nets)}
        # LR halving at epoch 1 (reference :447-456): G, im, st halved; se_optimizerD never decayed (quirk 13)
        assert tr.optimizerG.param_groups[0]['lr'] == pytest.approx(oc.g_lr / 4)
        assert tr.im_optimizerD.param_groups[0]['lr'] == pytest.approx(oc.d_lr / 4)
        assert tr.st_optimizerD.param_groups[0]['lr'] == pytest.approx(oc.d_lr / 4)
        assert tr.se_optimizerD.param_groups[0]['lr'] == pytest.approx(oc.d_lr)
        for opt, lr in ((tr.optimizerG, oc.g_lr / 4), (tr.se_optimizerD, oc.d_lr)):
            assert float(opt._hypers[0][0][1]) == pytest.approx(lr)           # the device-side scalar the kernel reads
            assert float(opt._hypers[0][0][0]) == 9.0                         # nine Adam steps
        assert getattr(tr.__dict__.get("_ng"), "captured", False), "train() did not reach the captured launch mode"
        # logging (reference :357-360 every step, :432-435 every 20 steps, :437-444 one sheet per epoch)
        rows = tr._logger.flush()
        std = [(k, s_) for k, v, s_ in rows if k == "st_D/loss"]
        assert [s_ for _, s_ in std] == list(range(9)), std                   # the story critic's scalars: EVERY step
        assert all(v == v for k, v, s_ in rows if k.startswith("st_D/"))
        assert sorted(s_ for k, v, s_ in rows if k == "G/loss") == [0, 3, 6]    # i % 20 == 0 of each epoch
        assert [(t, e) for t, shp, e in tr._logger.images] == [("pororo", 0), ("segment", 0), ("pororo", 1), ("segment", 1), ("pororo", 2), ("segment", 2)]
        assert all(len(shp) == 3 and shp[0] == 3 for t, shp, e in tr._logger.images)
        for e in range(3):
            assert os.path.exists(os.path.join(out_dir, "Image", "fake_samples_%d.txt" % e))
        model_dir = os.path.join(out_dir, "Model")
        for f in ("netG_epoch_0.pth", "netG_epoch_1.pth", "netG_epoch_2.pth", "netG_epoch_3.pth", "netD_im_epoch_last.pth",
                  "netD_st_epoch_last.pth", "netD_se_epoch_last.pth"):
            assert os.path.exists(os.path.join(model_dir, f)), f
        assert os.path.exists(os.path.join(out_dir, "model.py")) and os.path.exists(os.path.join(out_dir, "trainer.py"))
        for n in tr.nets:
            for t in list(n.parameters()) + list(n.buffers()):
                assert torch.isfinite(t).all()

        # ---- (2) the same nine steps driven by hand (train_step on the same batches, same seeds, LR halved by hand
        # after epochs 1 and 2): train() must land on the same weights -> batch order, image-loader wrap-around, LR timing
        ref_tr = T.GANTrainer(None, args, ratio=1.0)
        torch.manual_seed(11)
        ref_tr.setup()
        im_list, st_list = imageloader.batches, storyloader.batches
        dev = lambda b: {k: v.cuda() for k, v in b.items() if k != "text"}
        step = 0
        td = oc.text_dim
        for epoch in range(3):
            for s in st_list:
                ref_tr.train_step(dev(s), dev(im_list[step % len(im_list)]))
                step += 1
            # the epoch-end sample of the reference loop (trainer.py:437-444): train mode, no_grad, draws noise, moves BN statistics
            last = dev(st_list[-1])
            with torch.no_grad():
                ref_tr.nets[0].sample_videos(torch.cat((last["description"][:, :, :td], last["labels"]), 2), last["description"][:, :, :td], seg=True)
            if epoch >= 1:
                for opt in (ref_tr.optimizerG, ref_tr.st_optimizerD, ref_tr.im_optimizerD):
                    for g in opt.param_groups:
                        g['lr'] *= 0.5
                    opt.sync_lr()
        torch.cuda.synchronize()
        for k, n in zip("G im st se".split(), ref_tr.nets):
            for a, b in zip(init[k], n.parameters()):
                assert (a - b).abs().max().item() <= 2.2 * 9 * 4e-4, k       # same trajectory up to Adam's +-lr on round-off signs
        gw = torch.cat([p.flatten() for p in init["G"]])
        rw = torch.cat([p.detach().flatten() for p in ref_tr.nets[0].parameters()])
        assert (gw - rw).norm().item() <= 1e-3 * gw.norm().item()

        # ---- (3) wire format: the files hold exactly the reference's keys and load into the oracle (strict)
        from oracle.cpcsv_oracle import make_state
        st = make_state(oc)
        sd_g = torch.load(os.path.join(model_dir, "netG_epoch_3.pth"), map_location="cpu")
        assert set(sd_g) == set(gu.group(fx, "before/G"))
        st.netG.load_state_dict(sd_g, strict=True)
        st.netD_im.load_state_dict(torch.load(os.path.join(model_dir, "netD_im_epoch_last.pth"), map_location="cpu"), strict=True)
        st.netD_st.load_state_dict(torch.load(os.path.join(model_dir, "netD_st_epoch_last.pth"), map_location="cpu"), strict=True)
        st.netD_se.load_state_dict(torch.load(os.path.join(model_dir, "netD_se_epoch_last.pth"), map_location="cpu"), strict=True)
        bn = [v for k, v in sd_g.items() if k.endswith("num_batches_tracked")]
        assert bn and all(int(v) > 0 for v in bn)

        # ---- (4) resume (reference trainer.py:121-131): a new trainer with continue_ckpt loads these files
        res = T.GANTrainer(out_dir, types.SimpleNamespace(cfg_file=None, continue_ckpt="3"), ratio=1.0)
        rnets = res.load_network_stageI()
        for a, b in zip(rnets, tr.nets):
            for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
                assert ka == kb and torch.equal(va.cpu(), vb.cpu()), ka

        # ---- (5) eval-mode forward of the reloaded nets (BN running statistics, spectral norm frozen) vs the oracle
        stb, imb = gu.batches(fx)
        td = oc.text_dim
        st_motion = torch.cat((stb["description"][:, :, :td], stb["labels"]), 2)
        im_motion = torch.cat((imb["description"][:, :td], imb["labels"]), 1)
        tape = gu.noise_tape(fx)
        from oracle.cpcsv_oracle import NoiseTape
        for n in (st.netG, st.netD_im, st.netD_st, st.netD_se):
            n.eval()
        for n in rnets:
            n.eval()
        with torch.no_grad():
            onoise = NoiseTape(tape)
            _, ov, _, _, _, _, oseg = st.netG.sample_videos(st_motion, stb["description"][:, :, :td], seg=True, noise=onoise)
            _, oi, _, _, _, _, _ = st.netG.sample_images(im_motion, imb["content"][:, :, :td], seg=True, noise=onoise)
            pu.set_noise(rnets[0], pu.TapeSource(tape))
            _, pv, _, _, _, _, pseg = rnets[0].sample_videos(st_motion.cuda(), stb["description"][:, :, :td].cuda(), seg=True)
            _, pi, _, _, _, _, _ = rnets[0].sample_images(im_motion.cuda(), imb["content"][:, :, :td].cuda(), seg=True)
            assert gu.rel_err(pv.contiguous(), ov.contiguous()) < 2e-4
            assert gu.rel_err(pseg, oseg) < 2e-4 and gu.rel_err(pi, oi) < 2e-4
            # critics in eval mode: features and conditional logits; u/v must NOT advance
            u_before = rnets[1].encode_img[2].weight_u.clone()
            of = st.netD_im(imb["images"])
            pf = rnets[1](imb["images"].cuda())
            assert gu.rel_err(pf.float(), of) < 5e-4
            cond = torch.randn(imb["images"].shape[0], oc.critic_cond_dim, generator=torch.Generator().manual_seed(3))
            ol = st.netD_im.get_cond_logits(of, cond)
            pl = rnets[1].get_cond_logits(pf, cond.cuda())
            assert gu.rel_err(pl, ol) < 5e-4
            assert torch.equal(u_before, rnets[1].encode_img[2].weight_u)
            os_ = st.netD_st(stb["images"])
            ps = rnets[2](stb["images"].cuda())
            assert gu.rel_err(ps.float(), os_) < 5e-4
    finally:
        runtime.set_deterministic(was)
        from miscc.config import cfg as c
        c.TRAIN.FLAG = True
