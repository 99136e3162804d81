#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the real reference on CPU. TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs /root/reference). The reference's model.py,
cascade_model.py, layers.py and miscc/utils.py are imported unmodified with:
  * stub packages oracle/ref_shims/{easydict,torchvision}  (absent from the image),
  * torch.Tensor.cuda / nn.Module.cuda -> identity, cfg.CUDA=False  (model.py:238,55-58),
  * nn.parallel.data_parallel -> direct call  (miscc/utils.py:58-166).
trainer.py cannot be imported (tensorboardX, torchfile, pytorch_ssim missing), so its loop
body (trainer.py:252-416) is driven here against the imported reference nets and loss
functions. Fixtures hold DATA only: configs, weights, inputs, recorded noise, outputs.

    python oracle/gen_golden.py            # writes tests/golden/step_plain.npz, step_cascade.npz, ops.npz
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("CPCSV_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

THREADS = 4
torch.set_num_threads(THREADS)

# --- shims -------------------------------------------------------------------------------
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self


def _direct(module, inputs, device_ids=None, *a, **k):
    return module(*inputs) if isinstance(inputs, tuple) else module(inputs)


nn.parallel.data_parallel = _direct

from miscc.config import cfg as RCFG  # noqa: E402  (reference)
import miscc.utils as RU  # noqa: E402  (reference)

RCFG.CUDA = False

from oracle.cpcsv_oracle import NoiseTape, synthetic_batch, tiny_cfg  # noqa: E402


def apply_cfg(oc):
    """Push an OracleCfg into the reference's global cfg (keys of SURVEY §5 config row)."""
    RCFG.VIDEO_LEN = oc.video_len
    RCFG.LABEL_NUM = oc.label_num
    RCFG.TEXT.DIMENSION = oc.text_dim
    RCFG.GAN.CONDITION_DIM = oc.cond_dim
    RCFG.GAN.Z_DIM = oc.z_dim
    RCFG.GAN.DF_DIM = oc.df_dim
    RCFG.GAN.GF_DIM = oc.gf_dim
    RCFG.GAN.GF_SEG_DIM = oc.gf_seg_dim
    RCFG.SEGMENT_LEARNING = oc.segment_learning
    RCFG.SEGMENT_RATIO = oc.segment_ratio
    RCFG.IMAGE_RATIO = oc.image_ratio
    RCFG.RECONSTRUCT_LOSS = oc.reconstruct_loss
    RCFG.CASCADE_MODEL = oc.cascade
    RCFG.USE_SEQ_CONSISTENCY = False
    RCFG.TRAIN.COEFF.KL = oc.kl_coeff
    RCFG.TRAIN.IM_BATCH_SIZE = oc.im_batch
    RCFG.TRAIN.ST_BATCH_SIZE = oc.st_batch


def flat_state(prefix, net, out):
    for k, v in net.state_dict().items():
        out["%s/%s" % (prefix, k)] = v.detach().cpu().numpy().copy()


def summarise(t):
    t = t.detach().double().flatten()
    head = torch.zeros(8, dtype=torch.float64)
    head[:min(8, t.numel())] = t[:8]
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), float(t.numel())], head.numpy()])


def reference_step(oc, seed_w, seed_data, seed_noise, tag):
    """Drive trainer.py:252-416 against the imported reference nets."""
    apply_cfg(oc)
    import importlib
    mod = importlib.import_module("cascade_model" if oc.cascade else "model")
    torch.manual_seed(seed_w)                                   # main_pororo.py:53
    netG = mod.StoryGAN(oc.video_len)
    netG.apply(RU.weights_init)                                 # trainer.py:87-88
    netD_im = mod.STAGE1_D_IMG(); netD_im.apply(RU.weights_init)
    netD_st = mod.STAGE1_D_STY_V2(); netD_st.apply(RU.weights_init)
    netD_se = mod.STAGE1_D_SEG(); netD_se.apply(RU.weights_init)
    opt = lambda n, lr: torch.optim.Adam(n.parameters(), lr=lr, betas=(0.5, 0.999))   # trainer.py:212-220
    oG, oIm, oSt, oSe = opt(netG, oc.g_lr), opt(netD_im, oc.d_lr), opt(netD_st, oc.d_lr), opt(netD_se, oc.d_lr)

    fx = {}
    for p, n in (("G", netG), ("D_im", netD_im), ("D_st", netD_st), ("D_se", netD_se)):
        flat_state("before/" + p, n, fx)

    st_b, im_b = synthetic_batch(oc, seed=seed_data)
    for k, v in st_b.items():
        fx["batch/st/" + k] = v.numpy()
    for k, v in im_b.items():
        fx["batch/im/" + k] = v.numpy()

    td = oc.text_dim
    gpus = [0]
    im_real = im_b["images"]; im_labels = im_b["labels"]; se_real = im_b["images_seg"]
    im_motion = torch.cat((im_b["description"][:, :td], im_labels), 1)          # trainer.py:255,287
    im_content = im_b["content"][:, :, :td]                                     # :256
    st_real = st_b["images"]; st_labels = st_b["labels"]
    st_motion = torch.cat((st_b["description"][:, :, :td], st_labels), 2)       # :263,288
    st_content = st_b["description"][:, :, :td]                                 # :264
    one_im, zero_im = torch.ones(oc.im_batch), torch.zeros(oc.im_batch)         # :195-198
    one_st, zero_st = torch.ones(oc.st_batch), torch.zeros(oc.st_batch)

    torch.manual_seed(seed_noise)
    with torch.no_grad():                                                       # :295-300
        _, st_fake, _, _, c_mu, c_logvar, _ = netG.sample_videos(st_motion, st_content)
        _, im_fake, _, _, cim_mu, cim_logvar, se_fake = netG.sample_images(im_motion, im_content, seg=True)
    fx["nograd/st_fake"] = st_fake.contiguous().numpy(); fx["nograd/im_fake"] = im_fake.numpy()
    fx["nograd/se_fake"] = se_fake.numpy(); fx["nograd/c_mu"] = c_mu.numpy(); fx["nograd/c_logvar"] = c_logvar.numpy()
    fx["nograd/cim_mu"] = cim_mu.numpy(); fx["nograd/cim_logvar"] = cim_logvar.numpy()

    who = (st_labels.mean(1) > 0).type(torch.FloatTensor)                       # :303
    st_mu = torch.cat((c_mu, st_motion[:, :, :td].mean(1).squeeze(), who), 1)   # :304
    im_mu = torch.cat((im_motion, cim_mu), 1)                                   # :307
    fx["nograd/st_cond"] = st_mu.numpy(); fx["nograd/im_cond"] = im_mu.numpy()

    netD_im.zero_grad(); netD_st.zero_grad(); netD_se.zero_grad()               # :313-317
    se = RU.compute_discriminator_loss(netD_se, se_real, se_fake, one_im, zero_im, im_labels, im_mu, gpus)
    im = RU.compute_discriminator_loss(netD_im, im_real, im_fake, one_im, zero_im, im_labels, im_mu, gpus)
    st = RU.compute_discriminator_loss(netD_st, st_real, st_fake, one_st, zero_st, st_labels, st_mu, gpus)
    se[0].backward()
    for k, p in netD_se.named_parameters():
        fx["grad/D_se/" + k] = p.grad.numpy().copy()
    oSe.step()                                                                  # :334-335
    im[0].backward(); st[0].backward()                                          # :342-343
    for k, p in netD_im.named_parameters():
        fx["grad/D_im/" + k] = p.grad.numpy().copy()
    for k, p in netD_st.named_parameters():
        fx["grad/D_st/" + k] = p.grad.numpy().copy()
    oIm.step(); oSt.step()                                                      # :345-346
    sc = {}
    for nm, r in (("se_D", se), ("im_D", im), ("st_D", st)):
        sc[nm + "_loss"] = r[0].item(); sc[nm + "_real"] = r[1].item()
        sc[nm + "_wrong"] = r[2].item(); sc[nm + "_fake"] = r[3].item()
        if nm != "st_D":
            sc[nm + "_acc"] = float(r[4])

    netG.zero_grad()                                                            # :365
    v_lat, st_fake, _, _, c_mu, c_logvar, _ = netG.sample_videos(st_motion, st_content)
    i_lat, im_fake, _, _, cim_mu, cim_logvar, se_fake = netG.sample_images(im_motion, im_content, seg=True)
    fx["grad_pass/st_fake"] = st_fake.detach().contiguous().numpy()
    fx["grad_pass/im_fake"] = im_fake.detach().numpy(); fx["grad_pass/se_fake"] = se_fake.detach().numpy()
    mse = nn.MSELoss()
    extra = None
    if v_lat is not None:                                                       # :370-384
        (h1, h2, h3, h4), (g1, g2, g3, g4) = v_lat
        v_loss = mse(g1, h1) + mse(g2, h2) + mse(g3, h3) + mse(g4, h4)
        (h1, h2, h3, h4), (g1, g2, g3, g4) = i_lat
        i_loss = mse(g1, h1) + mse(g2, h2) + mse(g3, h3) + mse(g4, h4)
        r_img = netG.train_autoencoder(se_real); r_fake = netG.train_autoencoder(se_fake)
        rec = (mse(r_img, se_real) + mse(r_fake, se_fake)) / 2.0
        extra = v_loss + rec
        sc.update(video_latent=v_loss.item(), image_latent=i_loss.item(), reconstruct=rec.item())
    who = (st_labels.mean(1) > 0).type(torch.FloatTensor)                       # :386-389
    st_mu = torch.cat((c_mu, st_motion[:, :, :td].mean(1).squeeze(), who), 1)
    im_mu = torch.cat((im_motion, cim_mu), 1)
    se_g, se_acc, _ = RU.compute_generator_loss(netD_se, se_fake, se_real, one_im, im_labels, im_mu, gpus)
    im_g, im_acc, _ = RU.compute_generator_loss(netD_im, im_fake, im_real, one_im, im_labels, im_mu, gpus)
    st_g, st_acc, _ = RU.compute_generator_loss(netD_st, st_fake, st_real, one_st, st_labels, st_mu, gpus)
    im_kl = RU.KL_loss(cim_mu, cim_logvar); st_kl = RU.KL_loss(c_mu, c_logvar)  # :402-403
    total = im_g + im_kl * oc.kl_coeff + 1.0 * (se_g * oc.segment_ratio + st_g * oc.image_ratio
                                                + st_kl * oc.kl_coeff)           # :409-410
    if extra is not None:
        total = total + extra * oc.reconstruct_loss                              # :413
    total.backward()
    for k, p in netG.named_parameters():
        fx["grad/G/" + k] = p.grad.numpy().copy()
    oG.step()                                                                    # :416
    sc.update(G_loss=total.item(), im_G=im_g.item(), st_G=st_g.item(), se_G=se_g.item(),
              im_KL=im_kl.item(), st_KL=st_kl.item(), im_G_acc=float(im_acc), se_G_acc=float(se_acc),
              st_G_acc=float(st_acc))
    for k, v in sc.items():
        fx["scalar/" + k] = np.float64(v)
    for p, n in (("G", netG), ("D_im", netD_im), ("D_st", netD_st), ("D_se", netD_se)):
        for k, v in n.state_dict().items():
            fx["after/%s/%s" % (p, k)] = summarise(v)

    # record the noise tape: same seed, same draw order -> identical tensors (checked by the test:
    # the oracle fed this tape must reproduce the reference outputs)
    from oracle.cpcsv_oracle.nets import StoryGenerator, CascadeStoryGenerator
    shadow = (CascadeStoryGenerator if oc.cascade else StoryGenerator)(oc)   # built BEFORE seeding
    torch.manual_seed(seed_noise)
    tape = NoiseTape()
    with torch.no_grad():
        for _ in range(2):
            shadow.sample_videos(st_motion, st_content, noise=tape)
            shadow.sample_images(im_motion, im_content, noise=tape)
    for i, t in enumerate(tape.tape):
        fx["noise/%03d" % i] = t.numpy()

    fx["meta/cfg"] = np.array(repr(oc))
    fx["meta/seeds"] = np.array([seed_w, seed_data, seed_noise, THREADS])
    path = os.path.join(REPO, "tests", "golden", "step_%s.npz" % tag)
    np.savez_compressed(path, **fx)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6), "G_loss", sc["G_loss"])


def reference_ops():
    """Per-op micro goldens from the imported reference (layers.py, miscc/utils.py)."""
    from layers import DynamicFilterLayer1D  # reference
    fx = {}
    g = torch.Generator().manual_seed(7)
    sig = torch.randn(5, 3, 124, generator=g, requires_grad=True)
    taps = torch.randn(5, 1, 3, 21, generator=g, requires_grad=True)
    out = DynamicFilterLayer1D(21, pad=10)([sig, taps])                      # layers.py:69-80
    up = torch.randn(out.shape, generator=g)
    out.backward(up)
    fx.update({"dfl/sig": sig.detach().numpy(), "dfl/taps": taps.detach().numpy(), "dfl/out": out.detach().numpy(),
               "dfl/up": up.numpy(), "dfl/dsig": sig.grad.numpy(), "dfl/dtaps": taps.grad.numpy()})
    mu = torch.rand(6, 124, generator=g); lv = torch.rand(6, 124, generator=g)
    fx.update({"kl/mu": mu.numpy(), "kl/logvar": lv.numpy(), "kl/out": np.float64(RU.KL_loss(mu.clone(), lv.clone()).item())})
    logits = torch.randn(7, 9, generator=g); lab = (torch.rand(7, 9, generator=g) < 0.4).float(); lab[:, 0] = 1
    fx.update({"acc/logits": logits.numpy(), "acc/labels": lab.numpy(),
               "acc/out": np.float64(RU.get_multi_acc(logits.numpy(), lab.numpy()))})
    path = os.path.join(REPO, "tests", "golden", "ops.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path)


if __name__ == "__main__":
    base = tiny_cfg(cond_dim=12, gf_dim=4, gf_seg_dim=16, df_dim=8)
    reference_step(base, 0, 1, 1234, "plain")
    reference_step(base.but(cascade=True), 0, 1, 1234, "cascade")
    reference_ops()
