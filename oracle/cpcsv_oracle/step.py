"""One training step of GANTrainer.train, restated. TEST INFRASTRUCTURE ONLY.

Restates the loop body /root/reference/trainer.py:252-416 plus the optimiser set-up
(:212-220). trainer.py itself cannot be imported (tensorboardX/torchfile/pytorch_ssim
missing), so gen_golden.py drives the imported reference model/loss functions with an
equivalent body and this module is compared against that.
"""
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from .losses import critic_loss, generator_loss, kl_term
from .nets import (CascadeStoryGenerator, FrameCritic, SegCritic, StoryCritic, StoryGenerator,
                   init_like_reference)


class NoiseTape:
    """Noise source that either records every draw (CPU global RNG, reference draw order:
    CA eps -> GRU h0 noise -> per-step noise, model.py:56-58,319,315) or replays a tape."""

    def __init__(self, tape=None):
        self.replay = tape is not None
        self.tape = list(tape) if tape is not None else []
        self.pos = 0

    def __call__(self, shape):
        if self.replay:
            t = self.tape[self.pos]
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            self.pos += 1
            return t
        t = torch.empty(shape).normal_()
        self.tape.append(t)
        return t


@dataclass
class TrainState:
    cfg: object
    netG: torch.nn.Module
    netD_im: torch.nn.Module
    netD_st: torch.nn.Module
    netD_se: torch.nn.Module
    optG: torch.optim.Optimizer
    optD_im: torch.optim.Optimizer
    optD_st: torch.optim.Optimizer
    optD_se: torch.optim.Optimizer
    ratio: float = 1.0
    log: dict = field(default_factory=dict)


def make_state(cfg, seed=0):
    """load_network_stageI + optimisers, trainer.py:82-97,212-220."""
    torch.manual_seed(seed)
    g_cls = CascadeStoryGenerator if cfg.cascade else StoryGenerator
    netG = init_like_reference(g_cls(cfg))
    netD_im = init_like_reference(FrameCritic(cfg))
    netD_st = init_like_reference(StoryCritic(cfg))
    netD_se = init_like_reference(SegCritic(cfg)) if cfg.segment_learning else None
    adam = lambda net, lr: torch.optim.Adam(net.parameters(), lr=lr, betas=(0.5, 0.999))
    return TrainState(cfg, netG, netD_im, netD_st, netD_se,
                      adam(netG, cfg.g_lr), adam(netD_im, cfg.d_lr), adam(netD_st, cfg.d_lr),
                      adam(netD_se, cfg.d_lr) if netD_se is not None else None)


def synthetic_batch(cfg, seed=1, st=None, im=None):
    """Synthetic batch dicts with the schema trainer.py:254-274 reads."""
    g = torch.Generator().manual_seed(seed)
    st = st or cfg.st_batch
    im = im or cfg.im_batch
    t, d, nl = cfg.video_len, cfg.text_dim, cfg.label_num

    def labels(*shape):
        lab = (torch.rand(*shape, nl, generator=g) < 0.3).float()
        lab[..., 0] = torch.where(lab.sum(-1) == 0, torch.ones_like(lab[..., 0]), lab[..., 0])
        return lab

    story = {"images": torch.rand(st, 3, t, 64, 64, generator=g) * 2 - 1,
             "description": torch.randn(st, t, d, generator=g),
             "labels": labels(st, t)}
    image = {"images": torch.rand(im, 3, 64, 64, generator=g) * 2 - 1,
             "images_seg": torch.rand(im, 1, 64, 64, generator=g) * 2 - 1,
             "description": torch.randn(im, d, generator=g),
             "content": torch.randn(im, t, d + nl, generator=g),
             "labels": labels(im)}
    return story, image


def train_step(state, st_batch, im_batch, noise=None, before_step=None, shuffle=None):
    """trainer.py:252-416. Returns a dict of every scalar the reference logs.
    `shuffle` = losses.shuffle_plan(...) for the story critic's order-consistency head (cfg.use_seq_consistency).
    `before_step(name, net)` (optional) is called right before each optimiser step, after that net's gradients were
    recorded: the data-parallel emulation in tests/test_gpu_dist.py averages the replicas' gradients there."""
    cfg = state.cfg
    before_step = before_step or (lambda name, net: None)
    noise = noise or NoiseTape()
    G, D_im, D_st, D_se = state.netG, state.netD_im, state.netD_st, state.netD_se
    use_seg = cfg.segment_learning
    td = cfg.text_dim

    # (1) batch prep, trainer.py:254-288
    im_real = im_batch["images"]
    im_labels = im_batch["labels"]
    im_motion = torch.cat((im_batch["description"][:, :td], im_labels), 1)
    im_content = im_batch["content"][:, :, :td]
    st_real = st_batch["images"]
    st_labels = st_batch["labels"]
    st_text = st_batch["description"][:, :, :td]
    st_motion = torch.cat((st_text, st_labels), 2)
    st_content = st_text
    se_real = im_batch["images_seg"] if use_seg else None
    nim, nst = im_real.shape[0], st_real.shape[0]
    one_im, zero_im = torch.ones(nim), torch.zeros(nim)
    one_st, zero_st = torch.ones(nst), torch.zeros(nst)

    # (2) fakes without grad, modules stay in train mode, trainer.py:295-300
    with torch.no_grad():
        _, st_fake, _, _, c_mu, _, _ = G.sample_videos(st_motion, st_content, noise=noise)
        _, im_fake, _, _, cim_mu, _, se_fake = G.sample_images(im_motion, im_content, seg=True, noise=noise)

    who = (st_labels.mean(1) > 0).float()                                   # :303
    st_cond = torch.cat((c_mu, st_motion[:, :, :td].mean(1).squeeze(), who), 1)   # :304
    im_cond = torch.cat((im_motion, cim_mu), 1)                             # :307

    # (3) critic updates, trainer.py:313-346 (order: se step, then im/st backward, im/st step)
    D_im.zero_grad()
    D_st.zero_grad()
    out = {}
    if use_seg:
        D_se.zero_grad()
        se_err, se_r, se_w, se_f, se_acc, _ = critic_loss(D_se, se_real, se_fake, one_im, zero_im, im_labels, im_cond)
    im_err, im_r, im_w, im_f, im_acc, _ = critic_loss(D_im, im_real, im_fake, one_im, zero_im, im_labels, im_cond)
    st_err, st_r, st_w, st_f, _, st_cons = critic_loss(D_st, st_real, st_fake, one_st, zero_st, st_labels, st_cond,
                                                       shuffle=shuffle, consistency_ratio=cfg.consistency_ratio)
    if use_seg:
        se_err.backward()
        out["grads_D_se"] = _grads(D_se)
        before_step("D_se", D_se)
        state.optD_se.step()
        out.update(se_D_loss=se_err.item(), se_D_real=se_r.item(), se_D_wrong=se_w.item(),
                   se_D_fake=se_f.item(), se_D_acc=se_acc)
    im_err.backward()
    st_err.backward()
    out["grads_D_im"] = _grads(D_im)
    out["grads_D_st"] = _grads(D_st)
    before_step("D_im", D_im)
    before_step("D_st", D_st)
    state.optD_im.step()
    state.optD_st.step()
    out.update(im_D_loss=im_err.item(), im_D_real=im_r.item(), im_D_wrong=im_w.item(),
               im_D_fake=im_f.item(), im_D_acc=im_acc,
               st_D_loss=st_err.item(), st_D_real=st_r.item(), st_D_wrong=st_w.item(), st_D_fake=st_f.item(),
               st_D_consistency=st_cons)

    # (4) generator update, trainer.py:365-416
    G.zero_grad()
    v_lat, st_fake, _, _, c_mu, c_logvar, _ = G.sample_videos(st_motion, st_content, noise=noise)
    i_lat, im_fake, _, _, cim_mu, cim_logvar, se_fake = G.sample_images(im_motion, im_content, seg=use_seg, noise=noise)
    extra = None
    if v_lat is not None:                                                   # cascade, :370-384
        pair = lambda lat: sum(F.mse_loss(g, h) for h, g in zip(lat[0], lat[1]))
        video_latent = pair(v_lat)
        image_latent = pair(i_lat)
        rec_real = G.train_autoencoder(se_real)
        rec_fake = G.train_autoencoder(se_fake)
        reconstruct = (F.mse_loss(rec_real, se_real) + F.mse_loss(rec_fake, se_fake)) / 2.0
        extra = video_latent + reconstruct                                  # :413 (image_latent is logged only)
        out.update(video_latent=video_latent.item(), image_latent=image_latent.item(),
                   reconstruct=reconstruct.item())
    who = (st_labels.mean(1) > 0).float()
    st_cond = torch.cat((c_mu, st_motion[:, :, :td].mean(1).squeeze(), who), 1)
    im_cond = torch.cat((im_motion, cim_mu), 1)
    se_g, se_gacc = 0, 0
    if use_seg:
        se_g, se_gacc, _ = generator_loss(D_se, se_fake, se_real, one_im, im_labels, im_cond)
    im_g, im_gacc, _ = generator_loss(D_im, im_fake, im_real, one_im, im_labels, im_cond)
    st_g, st_gacc, st_gcons = generator_loss(D_st, st_fake, st_real, one_st, st_labels, st_cond,
                                             consistency_ratio=cfg.consistency_ratio)
    im_kl = kl_term(cim_mu, cim_logvar)
    st_kl = kl_term(c_mu, c_logvar)
    total = im_g + im_kl * cfg.kl_coeff + state.ratio * (
        se_g * cfg.segment_ratio + st_g * cfg.image_ratio + st_kl * cfg.kl_coeff)     # :409-410
    if extra is not None:
        total = total + extra * cfg.reconstruct_loss
    total.backward()
    out["grads_G"] = _grads(G)
    before_step("G", G)
    state.optG.step()
    out.update(G_loss=total.item(), im_G=im_g.item(), st_G=st_g.item(),
               se_G=(se_g.item() if use_seg else 0.0), im_KL=im_kl.item(), st_KL=st_kl.item(),
               im_G_acc=im_gacc, se_G_acc=se_gacc, st_G_acc=st_gacc, st_G_consistency=st_gcons)
    out["noise_tape"] = noise.tape
    return out


def _grads(net):
    return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
