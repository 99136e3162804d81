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

    python oracle/gen_golden.py            # writes tests/golden/{step_plain,step_cascade,step_clevr,
                                           #   steps3_plain,steps3_cascade,ops}.npz

Fixtures:
  step_<tag>.npz    ONE step, everything: weights before, batches, noise tape, no-grad outputs, every loss
                    scalar, every parameter gradient, post-step state summaries.
                    tags: plain, cascade (tiny Pororo-shaped dims), clevr (BASELINE config 1 dims: T=4, text 72,
                    labels 15, ST=2/IM=8 — datasets/clevr.py:24,38-41,104; tiny widths).
  steps3_<tag>.npz  K=3 consecutive steps (fresh batch and noise per step) starting from step_<tag>'s weights:
                    per step the batches, the noise tape, every scalar, summaries of every gradient and of the
                    whole post-step state (parameters after Adam, SN u/v, BN running statistics).
  fullwidth_{plain,cascade,bench}.npz  ONE step at cfg/final.yml widths, ST=3 / IM=9 (bench: ST=12 / IM=60) (python oracle/gen_golden.py --fullwidth):
                    weights by seed (reference init == oracle init, asserted), batch, noise, scalars, gradient / state summaries.
  eval_<tag>.npz    the EVAL-mode forward (inference.py:88-89) of the generator and the three critics on the state twelve
                    training steps leave behind: state, batch, noise tape, every output (reference_eval). tags: plain, cascade.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("CPCSV_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import dataclasses  # noqa: E402

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

from oracle.cpcsv_oracle import NoiseTape, clevr_cfg, synthetic_batch, tiny_cfg  # noqa: E402

NETS = ("G", "D_im", "D_st", "D_se")
ORDER_SEED = 4242

BIG = "seq_consisten_model."        # tensors under this prefix are stored as summaries only

from oracle import conditioning as COND  # noqa: E402


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
    RCFG.USE_SEQ_CONSISTENCY = oc.use_seq_consistency
    RCFG.CONSISTENCY_RATIO = oc.consistency_ratio
    RCFG.TRAIN.COEFF.KL = oc.kl_coeff
    RCFG.TRAIN.IM_BATCH_SIZE = oc.im_batch
    RCFG.TRAIN.ST_BATCH_SIZE = oc.st_batch


def summarise(t):
    t = t.detach().double().flatten()
    head = torch.zeros(8, dtype=torch.float64)
    head[:min(8, t.numel())] = t[:8]
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), float(t.numel())], head.numpy()])


class ReferenceRun:
    """The reference's nets + optimisers (trainer.py:84-97,212-220) and its loop body (trainer.py:252-416)."""

    def __init__(self, oc, seed_w):
        apply_cfg(oc)
        import importlib
        mod = importlib.import_module("cascade_model" if oc.cascade else "model")
        torch.manual_seed(seed_w)                                   # main_pororo.py:53
        self.oc = oc
        self.netG = mod.StoryGAN(oc.video_len)
        self.netG.apply(RU.weights_init)                            # trainer.py:87-88
        self.netD_im = mod.STAGE1_D_IMG(); self.netD_im.apply(RU.weights_init)
        self.netD_st = mod.STAGE1_D_STY_V2(); self.netD_st.apply(RU.weights_init)
        self.netD_se = mod.STAGE1_D_SEG(); self.netD_se.apply(RU.weights_init)
        opt = lambda n, lr: torch.optim.Adam(n.parameters(), lr=lr, betas=(0.5, 0.999))   # trainer.py:212-220
        self.oG, self.oIm = opt(self.netG, oc.g_lr), opt(self.netD_im, oc.d_lr)
        self.oSt, self.oSe = opt(self.netD_st, oc.d_lr), opt(self.netD_se, oc.d_lr)
        if oc.use_seq_consistency:
            # the order critic (VideoEncoder, 4.6 M parameters) is too large to commit: BOTH sides load the same
            # reproducible state instead (oracle.order_critic_state), only its seed and a checksum go into the fixture
            from oracle.cpcsv_oracle import order_critic_state
            res = self.netD_st.seq_consisten_model.load_state_dict(order_critic_state(ORDER_SEED), strict=True)
            assert not res.missing_keys and not res.unexpected_keys
        from oracle.cpcsv_oracle.nets import CascadeStoryGenerator, StoryGenerator
        # the oracle generator used ONLY to record the noise draws in the reference's order; built here, before any
        # noise seeding, so its own construction does not disturb the stream
        self.shadow = (CascadeStoryGenerator if oc.cascade else StoryGenerator)(oc)

    def nets(self):
        return zip(NETS, (self.netG, self.netD_im, self.netD_st, self.netD_se))

    def dump_state(self, fx, prefix, full):
        for p, n in self.nets():
            for k, v in n.state_dict().items():
                if full and k.startswith(BIG):
                    fx["%s_sum/%s/%s" % (prefix, p, k)] = summarise(v)
                else:
                    fx["%s/%s/%s" % (prefix, p, k)] = v.detach().cpu().numpy().copy() if full else summarise(v)

    def step(self, fx, pre, seed_data, seed_noise, full):
        """One iteration. `pre` prefixes every key ('' for the single-step fixtures); `full` stores whole gradient
        tensors and the no-grad outputs, otherwise 11-number summaries."""
        oc = self.oc
        netG, netD_im, netD_st, netD_se = self.netG, self.netD_im, self.netD_st, self.netD_se
        st_b, im_b = synthetic_batch(oc, seed=seed_data)
        for k, v in st_b.items():
            fx[pre + "batch/st/" + k] = v.numpy()
        for k, v in im_b.items():
            fx[pre + "batch/im/" + k] = v.numpy()

        def put_grads(tag, net):
            for k, p in net.named_parameters():
                if full and k.startswith(BIG):
                    fx["%sgradsum/%s/%s" % (pre, tag, k)] = summarise(p.grad)
                elif full:
                    fx["%sgrad/%s/%s" % (pre, tag, k)] = p.grad.numpy().copy()
                else:
                    fx["%sgradsum/%s/%s" % (pre, tag, k)] = summarise(p.grad)

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
        ng = {"st_fake": st_fake.contiguous(), "im_fake": im_fake, "se_fake": se_fake, "c_mu": c_mu, "c_logvar": c_logvar,
              "cim_mu": cim_mu, "cim_logvar": cim_logvar}
        who = (st_labels.mean(1) > 0).type(torch.FloatTensor)                       # :303
        st_mu = torch.cat((c_mu, st_motion[:, :, :td].mean(1).squeeze(), who), 1)   # :304
        im_mu = torch.cat((im_motion, cim_mu), 1)                                   # :307
        ng["st_cond"], ng["im_cond"] = st_mu, im_mu
        for k, v in ng.items():
            fx[pre + "nograd/" + k] = v.numpy() if full else summarise(v)

        netD_im.zero_grad(); netD_st.zero_grad(); netD_se.zero_grad()               # :313-317
        se = RU.compute_discriminator_loss(netD_se, se_real, se_fake, one_im, zero_im, im_labels, im_mu, gpus)
        im = RU.compute_discriminator_loss(netD_im, im_real, im_fake, one_im, zero_im, im_labels, im_mu, gpus)
        if oc.use_seq_consistency:
            # create_random_shuffle (miscc/utils.py:17-44) draws from the GLOBAL numpy and python generators: seed both,
            # record what it produced, and check that the oracle's restated plan gives the same tensor
            import random
            from oracle.cpcsv_oracle import apply_shuffle, shuffle_plan
            seen = {}
            orig_shuffle = RU.create_random_shuffle

            def spy(stories, random_rate=0.5):
                r = orig_shuffle(stories, random_rate)
                seen["imgs"], seen["labels"] = r[0].detach().clone(), r[1].detach().clone()
                return r
            RU.create_random_shuffle = spy
            np.random.seed(seed_noise)
            random.seed(seed_noise)
        st = RU.compute_discriminator_loss(netD_st, st_real, st_fake, one_st, zero_st, st_labels, st_mu, gpus)
        if oc.use_seq_consistency:
            RU.create_random_shuffle = orig_shuffle
            plan = shuffle_plan(oc.st_batch, oc.video_len, np.random.RandomState(seed_noise), random.Random(seed_noise))
            imgs, labels = apply_shuffle(st_real, plan)
            assert torch.equal(imgs, seen["imgs"]) and torch.equal(labels, seen["labels"]), "shuffle plan does not reproduce the reference"
            fx[pre + "shuffle/labels"] = np.array(plan[0]); fx[pre + "shuffle/src_story"] = np.array(plan[1])
            fx[pre + "shuffle/src_frame"] = np.array(plan[2])
        se[0].backward()
        put_grads("D_se", netD_se)
        self.oSe.step()                                                             # :334-335
        im[0].backward(); st[0].backward()                                          # :342-343
        put_grads("D_im", netD_im)
        put_grads("D_st", netD_st)
        self.oIm.step(); self.oSt.step()                                            # :345-346
        sc = {}
        for nm, r in (("se_D", se), ("im_D", im), ("st_D", st)):
            sc[nm + "_loss"] = r[0].item(); sc[nm + "_real"] = r[1].item()
            sc[nm + "_wrong"] = r[2].item(); sc[nm + "_fake"] = r[3].item()
            if nm != "st_D":
                sc[nm + "_acc"] = float(r[4])
        if oc.use_seq_consistency:
            sc["st_D_consistency"] = float(st[5])

        netG.zero_grad()                                                            # :365
        v_lat, st_fake, _, _, c_mu, c_logvar, _ = netG.sample_videos(st_motion, st_content)
        i_lat, im_fake, _, _, cim_mu, cim_logvar, se_fake = netG.sample_images(im_motion, im_content, seg=True)
        gp = {"st_fake": st_fake.detach().contiguous(), "im_fake": im_fake.detach(), "se_fake": se_fake.detach()}
        for k, v in gp.items():
            fx[pre + "grad_pass/" + k] = v.numpy() if full else summarise(v)
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
        st_g, st_acc, st_cons = RU.compute_generator_loss(netD_st, st_fake, st_real, one_st, st_labels, st_mu, gpus)
        if oc.use_seq_consistency:
            sc["st_G_consistency"] = float(st_cons)
        im_kl = RU.KL_loss(cim_mu, cim_logvar); st_kl = RU.KL_loss(c_mu, c_logvar)  # :402-403
        total = im_g + im_kl * oc.kl_coeff + 1.0 * (se_g * oc.segment_ratio + st_g * oc.image_ratio
                                                    + st_kl * oc.kl_coeff)           # :409-410
        if extra is not None:
            total = total + extra * oc.reconstruct_loss                              # :413
        total.backward()
        put_grads("G", netG)
        self.oG.step()                                                               # :416
        sc.update(G_loss=total.item(), im_G=im_g.item(), st_G=st_g.item(), se_G=se_g.item(),
                  im_KL=im_kl.item(), st_KL=st_kl.item(), im_G_acc=float(im_acc), se_G_acc=float(se_acc),
                  st_G_acc=float(st_acc))
        for k, v in sc.items():
            fx[pre + "scalar/" + k] = np.float64(v)
        self.dump_state(fx, pre + "after", full=False)

        # the noise tape: same seed, same draw order -> identical tensors (checked by the test: the oracle fed
        # this tape must reproduce the reference outputs). Nothing else in the step draws random numbers.
        torch.manual_seed(seed_noise)
        tape = NoiseTape()
        with torch.no_grad():
            for _ in range(2):
                self.shadow.sample_videos(st_motion, st_content, noise=tape)
                self.shadow.sample_images(im_motion, im_content, noise=tape)
        for i, t in enumerate(tape.tape):
            fx["%snoise/%03d" % (pre, i)] = t.numpy()
        return sc


def cfg_json(oc):
    return np.array(json.dumps(dataclasses.asdict(oc), sort_keys=True))


def save(fx, name):
    path = os.path.join(REPO, "tests", "golden", name)
    np.savez_compressed(path, **fx)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


def oracle_state_from(run):
    """An oracle TrainState carrying the reference run's current weights and buffers (fresh Adam state: call before any step)."""
    from oracle.cpcsv_oracle import make_state
    st = make_state(run.oc)
    for (_, ref), net in zip(run.nets(), (st.netG, st.netD_im, st.netD_st, st.netD_se)):
        res = net.load_state_dict(ref.state_dict(), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    return st


def shuffle_for(oc, seed_noise):
    if not oc.use_seq_consistency:
        return None
    import random
    from oracle.cpcsv_oracle import shuffle_plan
    return shuffle_plan(oc.st_batch, oc.video_len, np.random.RandomState(seed_noise), random.Random(seed_noise))


def conditioning(oc, st, seed_data, seed_noise, steps):
    """oracle/conditioning.py over `steps` consecutive steps (step k: seeds + k) from the oracle state `st` (advanced in place):
    [(safety of the sensitive layers, flips in them, worst row, safety of all layers, flips in all)] per step."""
    out = []
    for k in range(steps):
        stb, imb = synthetic_batch(oc, seed=seed_data + k)
        torch.manual_seed(seed_noise + k)
        rows, _ = COND.kink_safety(st, stb, imb, shuffle=shuffle_for(oc, seed_noise + k))
        out.append(COND.summary(rows))
    return out


def put_conditioning(fx, cond, tag):
    fx["meta/kink_safety"] = np.array([c[0] for c in cond])             # per step, sensitive layers (n < SENSITIVE_NUMEL)
    fx["meta/kink_flips"] = np.array([c[1] for c in cond])
    fx["meta/kink_safety_all"] = np.array([c[3] for c in cond])
    fx["meta/kink_worst"] = np.array(["%s n=%d" % (c[2][0], c[2][1]) for c in cond])
    fx["meta/kink_rule"] = np.array([COND.SENSITIVE_NUMEL, SEARCH_TRIES])
    for k, c in enumerate(cond):
        print("   %s step %d: kink safety %.1f round-offs (%s, %d elements), flips %d; all layers %.2f / %d"
              % (tag, k, c[0], c[2][0], c[2][1], c[1], c[3], c[4]))
    assert all(c[1] == 0 for c in cond), "fp32 and fp64 disagree about a mask in a sensitive layer"


def step3_cfg(oc):
    """The three-step fixtures run ST=4 / IM=6 (the single-step ones ST=3 / IM=4): BatchNorm1d over three rows has an inverse
    standard deviation that amplifies round-off ~3x more than over four, and the state after a step feeds the next one."""
    return oc.but(st_batch=4, im_batch=6)


def reference_steps(oc, seed_w, seed_data, seed_noise, tag, k3=True, seeds3=None):
    """step_<tag>.npz from a fresh run, and (k3) steps3_<tag>.npz from ANOTHER fresh run with the same weight seed, its own
    batch sizes (step3_cfg) and its own data / noise seeds `seeds3` (step k uses seed + k)."""
    run = ReferenceRun(oc, seed_w)
    cond = conditioning(oc, oracle_state_from(run), seed_data, seed_noise, 1)
    fx = {}
    run.dump_state(fx, "before", full=True)
    sc = run.step(fx, "", seed_data, seed_noise, full=True)
    put_conditioning(fx, cond, tag)
    fx["meta/cfg_json"] = cfg_json(oc)
    fx["meta/seeds"] = np.array([seed_w, seed_data, seed_noise, THREADS])
    if oc.use_seq_consistency:
        fx["meta/order_seed"] = np.array(ORDER_SEED)
    save(fx, "step_%s.npz" % tag)
    print("   G_loss", sc["G_loss"])
    if not k3:
        return
    oc3 = step3_cfg(oc)
    sd3, sn3 = seeds3
    run = ReferenceRun(oc3, seed_w)
    check = {}
    run.dump_state(check, "before", full=True)
    assert all(np.array_equal(check[k], fx[k]) for k in check), "weight seed did not reproduce the weights"
    fx3 = {"meta/cfg_json": cfg_json(oc3), "meta/weights_from": np.array("step_%s.npz" % tag),
           "meta/seeds": np.array([seed_w, sd3, sn3, THREADS]), "meta/steps": np.array(3)}
    put_conditioning(fx3, conditioning(oc3, oracle_state_from(run), sd3, sn3, 3), tag + "3")
    for k in range(3):
        sck = run.step(fx3, "s%d/" % k, sd3 + k, sn3 + k, full=False)
        print("   step %d G_loss %.6f im_D %.6f st_D %.6f" % (k, sck["G_loss"], sck["im_D_loss"], sck["st_D_loss"]))
    save(fx3, "steps3_%s.npz" % tag)


EVAL_AFTER_STEPS = 12      # (after ONE step the running statistics are still 90 % their initial 0 / 1 and the eval-mode segmentation
                           #  decoder is dead: its output is exactly 0 - nothing to compare)


def reference_eval(oc, seed_w, seed_data, seed_noise, tag):
    """eval_<tag>.npz: the reference's EVAL-mode forward (inference.py:88-89 `netG.eval()` + `torch.no_grad()`, then
    sample_videos / sample_images as its StoryGANDataset calls them) and the three critics' eval-mode features / logits, on the
    state EVAL_AFTER_STEPS training steps leave behind (BatchNorm running statistics, spectral-norm u / v and every weight have
    moved off their initial values). Holds that whole state, the batch, the recorded noise, the outputs, and u / v after the eval calls
    (torch's spectral_norm does not iterate in eval mode: they must be the stored ones)."""
    run = ReferenceRun(oc, seed_w)
    for k in range(EVAL_AFTER_STEPS):
        run.step({}, "", seed_data + k, seed_noise + k, full=False)
    fx = {}
    run.dump_state(fx, "state", full=True)
    st_b, im_b = synthetic_batch(oc, seed=seed_data + 7)
    for k, v in st_b.items():
        fx["batch/st/" + k] = v.numpy()
    for k, v in im_b.items():
        fx["batch/im/" + k] = v.numpy()
    td = oc.text_dim
    im_motion = torch.cat((im_b["description"][:, :td], im_b["labels"]), 1)
    st_motion = torch.cat((st_b["description"][:, :, :td], st_b["labels"]), 2)
    st_content, im_content = st_b["description"][:, :, :td], im_b["content"][:, :, :td]
    for _, n in run.nets():
        n.eval()
    torch.manual_seed(seed_noise + 7)
    with torch.no_grad():
        _, st_fake, _, _, c_mu, c_logvar, st_seg = run.netG.sample_videos(st_motion, st_content, seg=True)
        _, im_fake, _, _, cim_mu, cim_logvar, se_fake = run.netG.sample_images(im_motion, im_content, seg=True)
        out = {"st_fake": st_fake.contiguous(), "st_seg": st_seg.contiguous(), "im_fake": im_fake, "se_fake": se_fake, "c_mu": c_mu,
               "c_logvar": c_logvar, "cim_mu": cim_mu, "cim_logvar": cim_logvar}
        who = (st_b["labels"].mean(1) > 0).type(torch.FloatTensor)
        st_mu = torch.cat((c_mu, st_motion[:, :, :td].mean(1).squeeze(), who), 1)
        im_mu = torch.cat((im_motion, cim_mu), 1)
        out["st_cond"], out["im_cond"] = st_mu, im_mu
        for nm, net, imgs, cond in (("D_im", run.netD_im, im_b["images"], im_mu), ("D_se", run.netD_se, im_b["images_seg"], im_mu),
                                    ("D_st", run.netD_st, st_b["images"], st_mu)):
            feats = net(imgs)
            out[nm + "_feats"] = feats
            out[nm + "_logits"] = net.get_cond_logits(feats, cond)
            if net.cate_classify is not None:
                out[nm + "_cate"] = net.cate_classify(feats)
    for k, v in out.items():
        fx["eval/" + k] = v.numpy()
    run.dump_state(fx, "after_eval", full=False)           # summaries: u / v / running statistics must not have moved
    torch.manual_seed(seed_noise + 7)
    tape = NoiseTape()
    run.shadow.eval()
    with torch.no_grad():
        run.shadow.sample_videos(st_motion, st_content, noise=tape)
        run.shadow.sample_images(im_motion, im_content, noise=tape)
    for i, t in enumerate(tape.tape):
        fx["noise/%03d" % i] = t.numpy()
    fx["meta/cfg_json"] = cfg_json(oc)
    fx["meta/seeds"] = np.array([seed_w, seed_data, seed_noise, THREADS])
    save(fx, "eval_%s.npz" % tag)


def reference_fullwidth(name="fullwidth_plain", st=3, im=9, cascade=False, seed_w=0, seed_data=1, seed_noise=5, keep_batch=True, **dims):
    """<name>.npz: ONE step of the reference at cfg/final.yml WIDTHS (ngf 2048, seg 1024, ndf 124, text 356, T=5) - the steps
    tests/test_gpu_fullsize.py evaluates with the oracle (fp32 and fp64) and the product: fullwidth_plain / fullwidth_cascade at ST=3 /
    IM=9, fullwidth_bench at the BENCHMARKED batch ST=12 / IM=60, fullwidth_clevr at the CLEVR dimensions (T=4, text 72, labels 15, ST=2 / IM=8). The 158 M weights are not stored: the reference built under
    torch.manual_seed(seed_w) and oracle.make_state(cfg, seed_w) produce the same tensors bit for bit (asserted here on every
    tensor; fixture meta/weights_sum holds their checksum), so both sides of the GPU test rebuild them from the seed. Stored: the
    batch (keep_batch=False: its checksum only - the bench batch is 7 MB of noise that synthetic_batch(seed) reproduces), the
    recorded noise, every scalar, 11-number summaries of every gradient, of the no-grad outputs and of the post-step state."""
    from oracle.cpcsv_oracle import make_state, pororo_cfg
    oc = pororo_cfg(st_batch=st, im_batch=im, cascade=cascade, **dims)
    run = ReferenceRun(oc, seed_w)
    ost = make_state(oc, seed=seed_w)
    total = 0.0
    for (nm, ref), net in zip(run.nets(), (ost.netG, ost.netD_im, ost.netD_st, ost.netD_se)):
        a, b = ref.state_dict(), net.state_dict()
        assert list(a) == list(b), nm
        for k in a:
            assert torch.equal(a[k], b[k]), (nm, k)
            total += float(a[k].double().abs().sum())
    del ost
    fx = {}
    sc = run.step(fx, "", seed_data, seed_noise, full=False)
    if not keep_batch:
        for k in [k for k in fx if k.startswith("batch/")]:
            fx["batchsum/" + k[len("batch/"):]] = summarise(torch.from_numpy(fx.pop(k)))
    fx["meta/cfg_json"] = cfg_json(oc)
    fx["meta/seeds"] = np.array([seed_w, seed_data, seed_noise, torch.get_num_threads()])
    fx["meta/weights_sum"] = np.float64(total)
    save(fx, name + ".npz")
    print("   G_loss", sc["G_loss"], "im_D", sc["im_D_loss"], "st_D", sc["st_D_loss"])


SEARCH_TRIES = 64


def search(oc, seed_w, steps, tries=SEARCH_TRIES):
    """Seed search: candidate i = (data seed 1 + 10 i, noise seed 1234 + 10 i), i < tries; a candidate's figure is the
    smallest kink safety of its sensitive layers over `steps` consecutive steps (0 if fp32 and fp64 disagree about a mask
    there). Returns the best candidate. Deterministic on one host: SEEDS below is what it returned in the build container;
    `python oracle/gen_golden.py --search` repeats it."""
    run = ReferenceRun(oc, seed_w)
    base = oracle_state_from(run)
    import copy
    best = (-1.0, None)
    for i in range(tries):
        sd, sn = 1 + 10 * i, 1234 + 10 * i
        st = copy.deepcopy(base)
        m = 1e30
        for k in range(steps):
            c = conditioning(oc, st, sd + k, sn + k, 1)[0]
            m = min(m, c[0] if c[1] == 0 else 0.0)
            if m <= best[0] and not os.environ.get("CPCSV_SEARCH_ALL"):
                break
        if os.environ.get("CPCSV_SEARCH_ALL"):
            print("   candidate %3d seeds (%d, %d): kink safety %.2f" % (i, sd, sn, m), flush=True)
        if m > best[0]:
            best = (m, (sd, sn))
            print("   candidate %3d seeds (%d, %d): kink safety %.2f  <- best so far" % (i, sd, sn, m), flush=True)
    return best


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
    save(fx, "ops.npz")


# (data seed, noise seed) per fixture, as returned by search() (python oracle/gen_golden.py --search re-derives them)
# kink safety reached (sensitive layers, smallest over the fixture's steps, best of SEARCH_TRIES = 64 candidates each): plain 5.0,
# plain3 1.4, cascade 2.1, cascade3 0.34 (see below), clevr 3.8, seq 2.1 round-offs - an arbitrary seed pair has 0.0 .. 1.5 (rounds 1-5 used
# (1, 1234): 0.00 in every one of the six searches). What the search cannot remove - every step keeps a few elements within a few
# round-offs of their kink - the lock-step tests RESOLVE (tests/parity_util.py resolve_kinks), they do not widen a band for it.
# cascade3: the search's best pair is (131, 1364) (0.65); its three FREE-RUNNING steps decorrelate from the reference's record faster
# than tests/test_oracle_vs_golden.py allows (generator gradient 2.8e-3 / 7.6e-2 / 0.30 at steps 0 / 1 / 2: Adam's lr * sign(g) on
# round-off-sized gradients, not a kink - conditioning.match_kink_sides over the 100 closest elements explains none of it), so the
# runner-up (221, 1454) (0.34; 2.9e-5 / 7.8e-3 / 0.10) is taken: a fixture has to pin the oracle to the reference first.
SEEDS = {"plain": (601, 1834), "plain3": (311, 1544), "cascade": (501, 1734), "cascade3": (221, 1454), "clevr": (571, 1804),
         "seq": (491, 1724)}


def configs():
    base = tiny_cfg(cond_dim=12, gf_dim=4, gf_seg_dim=16, df_dim=8)
    return {"plain": base, "cascade": base.but(cascade=True),
            # BASELINE config 1 dims (CLEVR: T=4, text 72, labels 15, ST=2/IM=8) at tiny widths
            "clevr": clevr_cfg(cond_dim=12, gf_dim=4, gf_seg_dim=16, df_dim=8),
            # the optional order-consistency critic (USE_SEQ_CONSISTENCY): VideoEncoder + create_random_shuffle
            "seq": base.but(use_seq_consistency=True, st_batch=6, im_batch=6)}


if __name__ == "__main__":
    cfgs = configs()
    if "--fullwidth" in sys.argv:
        torch.set_num_threads(8)
        reference_fullwidth("fullwidth_plain")
        reference_fullwidth("fullwidth_cascade", cascade=True)
        reference_fullwidth("fullwidth_bench", st=12, im=60, keep_batch=False)
        # BASELINE config 1 (CLEVR: T=4, text 72, labels 15, ST=2 / IM=8 - datasets/clevr.py:24,38-41,104) at the full widths
        reference_fullwidth("fullwidth_clevr", st=2, im=8, video_len=4, text_dim=72, label_num=15)
        sys.exit(0)
    if "--eval-only" in sys.argv:
        for tag in ("plain", "cascade"):
            reference_eval(cfgs[tag], 0, SEEDS[tag][0], SEEDS[tag][1], tag)
        sys.exit(0)
    if "--search" in sys.argv:
        only = sys.argv[sys.argv.index("--search") + 1:]
        for tag, oc in cfgs.items():
            if not only or tag in only:
                print("search", tag, flush=True)
                print("  ", tag, search(oc, 0, 1))
            if tag in ("plain", "cascade") and (not only or tag + "3" in only):
                print("search", tag + "3", flush=True)
                print("  ", tag + "3", search(step3_cfg(oc), 0, 3))
        sys.exit(0)
    for tag, oc in cfgs.items():
        sd, sn = SEEDS[tag]
        reference_steps(oc, 0, sd, sn, tag, k3=tag in ("plain", "cascade"), seeds3=SEEDS.get(tag + "3"))
    for tag in ("plain", "cascade"):
        reference_eval(cfgs[tag], 0, SEEDS[tag][0], SEEDS[tag][1], tag)
    reference_ops()
