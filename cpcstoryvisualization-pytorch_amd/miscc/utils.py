"""miscc/utils.py — loss glue of the CP-CSV step on the HIP path.

Drop-in for the hot-path part of the reference's miscc/utils.py:48-201,313-338: same function
names, argument order and return tuples. `gpus` is accepted and ignored: the reference fans the
critics out with single-process nn.parallel.data_parallel (25 broadcast/scatter/gather round trips
per step); here it is one process per GPU and gradients are exchanged once per optimiser
(cpcsv.dist). The sample dumps (reference :229-311,343-400) are CPU-side I/O restated at the end of this file
(torchvision's make_grid written out: torchvision is not part of this image).
"""
import os

import numpy as np
import torch

from cpcsv import functional as F
from miscc.config import cfg


shuffle_plan_source = None      # tests: callable(n_stories, video_len) -> (labels, src_story, src_frame), replaces the draws


def shuffle_plan(b, t, random_rate=0.5):
    """The DECISIONS of create_random_shuffle (reference miscc/utils.py:17-44), made on the host with the reference's
    generators in the reference's draw order (numpy: coin flip, re-shuffles; python `random`: permutation, donor, slot), so
    seeding both reproduces the reference: (order labels, source story per frame slot, source frame per frame slot)."""
    import random
    if shuffle_plan_source is not None:
        return shuffle_plan_source(b, t)
    labels, ss, sf = [], [], []
    for idx in range(b):
        label = 1 if random_rate > np.random.random() else 0
        row_s, row_f = [idx] * t, list(range(t))
        if label == 1:
            seq = random.sample(range(t), t)
            while bool((np.diff(seq) >= 0).all()):
                np.random.shuffle(seq)
            row_f = list(seq)
            donor = random.randint(0, b - 1)
            if donor != idx:
                slot = random.sample(range(t), 1)[0]
                row_s[slot], row_f[slot] = donor, slot
        labels.append(label)
        ss.append(row_s)
        sf.append(row_f)
    return labels, ss, sf


class ShufflePlanBuffers:
    """Persistent device tensors holding the current step's shuffle plan. The trainer refreshes them (tiny host->device
    copies, outside any captured graph) right before the story critic's update; create_random_shuffle then only GATHERS on
    the device through them - no host decision inside the critic's pass, so that pass can be captured as a HIP graph and
    replayed with a new plan every step."""

    SLOTS = 4          # pinned staging pairs in rotation (the host runs a few steps ahead of the GPU)

    def __init__(self, b, t, device):
        self.b, self.t = b, t
        self.src = torch.zeros(b * t, dtype=torch.long, device=device)          # flat frame index story*t + frame
        self.labels = torch.zeros(b, dtype=torch.float32, device=device)
        pin = device.type == "cuda" if isinstance(device, torch.device) else str(device).startswith("cuda")
        mk = lambda n, dt: torch.zeros(n, dtype=dt).pin_memory() if pin else torch.zeros(n, dtype=dt)
        # [host src, host labels, event recorded behind the copies that read them]: the train loop does not sync per step, so
        # with ONE staging pair the host could rewrite it for step N+1 while step N's queued copy has not read it yet
        # (src and labels torn across two plans). A slot is rewritten only after its last copy has completed.
        self._slots = [[mk(b * t, torch.long), mk(b, torch.float32), None] for _ in range(self.SLOTS)]
        self._next = 0
        self.armed = False

    def refresh(self, random_rate=0.5):
        labels, ss, sf = shuffle_plan(self.b, self.t, random_rate)
        slot = self._slots[self._next]
        self._next = (self._next + 1) % self.SLOTS
        hs, hl, ev = slot
        if ev is not None:
            ev.synchronize()                 # (SLOTS steps old: done long ago unless the host is that far ahead)
        hs.copy_(torch.tensor(ss).reshape(-1) * self.t + torch.tensor(sf).reshape(-1))
        hl.copy_(torch.tensor(labels, dtype=torch.float32))
        self.src.copy_(hs, non_blocking=True)
        self.labels.copy_(hl, non_blocking=True)
        if self.src.is_cuda:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ShufflePlanBuffers.refresh() inside a graph capture: the plan would be drawn once and frozen "
                                   "into the graph; refresh before the replay (GANTrainer.train_step_graphed does)")
            slot[2] = torch.cuda.Event()
            slot[2].record()
        self.armed = True


shuffle_buffers = None          # set by the trainer for the story critic's update (ShufflePlanBuffers), else None


def create_random_shuffle(stories, random_rate=0.5):
    """reference miscc/utils.py:17-44: every story is, with probability `random_rate`, frame-shuffled (never left
    sorted) and - unless the randomly chosen donor is the story itself - gets ONE frame slot overwritten by the donor
    story's frame of that slot. The decisions come from shuffle_plan (host, the reference's generators); the frames
    themselves are gathered on the device (the reference round-trips the batch through the CPU).
    Returns (shuffled stories (B,C,T,H,W), order labels (B,) float)."""
    b, c, t = stories.shape[0], stories.shape[1], stories.shape[2]
    buf = shuffle_buffers
    if buf is not None and buf.armed and (buf.b, buf.t) == (b, t) and buf.src.device == stories.device:
        src, labels = buf.src, buf.labels                       # refreshed by the trainer for this step
    else:
        lab, ss, sf = shuffle_plan(b, t, random_rate)
        dev = stories.device
        src = (torch.tensor(ss, device=dev).reshape(-1) * t + torch.tensor(sf, device=dev).reshape(-1))
        labels = torch.tensor(lab, dtype=torch.float32, device=dev)
    frames = stories.permute(0, 2, 1, 3, 4).reshape(b * t, c, stories.shape[3], stories.shape[4]).index_select(0, src)
    shuffled = frames.view(b, t, c, stories.shape[3], stories.shape[4]).permute(0, 2, 1, 3, 4)
    return shuffled, labels


def _bce(prob, target):
    return F.BceFn.apply(prob, target)


def _mlsm(logits4d, labels):
    """(MultiLabelSoftMarginLoss, get_multi_acc) of the category logits against the labels, one launch (the accuracy is a
    detached device scalar: no host round trip, reference miscc/utils.py:108,153 goes through numpy)."""
    logits = logits4d.reshape(logits4d.shape[0], -1)          # the reference's .squeeze(): (N,C,1,1) -> (N,C)
    return F.MlsmFn.apply(logits, labels.float(), logits.shape[1])


def _plus(total, coeff, term):
    """total + coeff * term without a multiply launch for coeff == 1"""
    return total + (term if coeff == 1.0 else coeff * term)


def multi_acc_device(logits, labels):
    """get_multi_acc without the device->host round trip: a 0-dim device tensor."""
    hit = ((labels == 1) & (logits >= 0)).sum()               # sigmoid(x) >= .5  <=>  x >= 0
    return hit.float() / labels.sum().float()


def compute_discriminator_loss(netD, real_imgs, fake_imgs, real_labels, fake_labels, real_catelabels, conditions, gpus,
                               real_features=None):
    """reference miscc/utils.py:48-123 (conditional branch). Returns
    (errD, errD_real, errD_wrong, errD_fake, acc, consistency_loss_val).
    `real_features` (extension): netD(real_imgs) computed earlier by the caller - it depends on nothing the generator
    produces, so the trainer runs it while the generator is still making the fakes."""
    batch_size = real_imgs.size(0)
    fake = fake_imgs.detach()
    if conditions is None:
        raise NotImplementedError("unconditional critics (reference :56-66) are never built by trainer.py")
    cond = conditions.detach()
    if (real_features is None and BATCH_PASSES and batch_size > 1 and fake.shape == real_imgs.shape and netD.training
            and netD.get_uncond_logits is None and hasattr(netD, "encode_pair")):
        return _discriminator_loss_batched(netD, real_imgs, fake, real_labels, fake_labels, real_catelabels, cond)
    if real_features is None:
        real_features = netD(real_imgs)                                        # :70
    fake_features = netD(fake)                                                 # :71
    errD_real = _bce(netD.get_cond_logits(real_features, cond), real_labels)   # :74-76
    wrong_logits = netD.get_cond_logits(real_features[:(batch_size - 1)], cond[1:])      # :78-79
    errD_wrong = _bce(wrong_logits, fake_labels[1:])                           # :80
    errD_fake = _bce(netD.get_cond_logits(fake_features, cond), fake_labels)   # :82-84
    if netD.get_uncond_logits is not None:
        raise NotImplementedError("get_uncond_logits is always None in the reference models (model.py:517)")
    errD = errD_real + (errD_fake + errD_wrong) * 0.5                          # :101
    acc = 0
    if netD.cate_classify is not None:                                         # :104-108
        cate_loss, acc = _mlsm(netD.cate_classify(real_features), real_catelabels)
        errD = errD + cate_loss
    consistency = 0
    if netD.seq_consisten_model is not None:                                   # :110-122
        shuffled, order_labels = create_random_shuffle(real_imgs)
        order_logits = netD.seq_consisten_model(shuffled)                      # (B, 1)
        consistency, _ = F.MlsmFn.apply(order_logits, order_labels.unsqueeze(-1), 1)    # == nn.BCEWithLogitsLoss, one class
        errD = _plus(errD, cfg.CONSISTENCY_RATIO, consistency)
        consistency = consistency.detach()
    return errD, errD_real.detach(), errD_wrong.detach(), errD_fake.detach(), acc, consistency


BATCH_PASSES = os.environ.get("CPCSV_BATCH_PASSES", "1") != "0"


def _discriminator_loss_batched(netD, real_imgs, fake, real_labels, fake_labels, real_catelabels, cond):
    """compute_discriminator_loss with the reference's five critic calls (tower(real), tower(fake), head(real), head(wrong),
    head(fake); miscc/utils.py:70-84) run as TWO passes: the tower over [real | fake] and the head over [real | wrong |
    fake] (model._Critic.encode_pair / D_GET_LOGITS.forward_triplet). Every call keeps its own BatchNorm batch and
    spectral-norm iteration, in the reference's order; only the launches are shared."""
    n = real_imgs.size(0)
    feats = netD.encode_pair(real_imgs, fake)                                  # :70-71, rows [0,n) real, [n,2n) fake
    probs = netD.get_cond_logits.forward_triplet(feats, cond)                  # :74-84
    # [real | wrong | fake] targets: the labels are the trainer's persistent ones / zeros vectors, so the concatenation is built
    # once per critic (the cache lives and dies with the module; rebuilt if the caller hands other label tensors)
    cache = netD.__dict__.setdefault("_bce_targets", {})
    key = (real_labels.data_ptr(), fake_labels.data_ptr(), real_labels._version, fake_labels._version, n)
    target = cache.get(key)
    if target is None:
        if torch.cuda.is_current_stream_capturing():
            target = torch.cat((real_labels[:n], fake_labels[1:n], fake_labels[:n])).float()     # graph-pool memory: not cached
        else:
            cache.clear()
            target = cache[key] = torch.cat((real_labels[:n], fake_labels[1:n], fake_labels[:n])).float()
    errD, parts = F.BceGroupsFn.apply(probs, target, (n, n - 1, n), (1.0, 0.5, 0.5))    # :76,80,84,101
    acc = 0
    if netD.cate_classify is not None:                                         # :104-108
        cate_loss, acc = _mlsm(netD.cate_classify(feats[:n]), real_catelabels)
        errD = errD + cate_loss
    consistency = 0
    if netD.seq_consisten_model is not None:                                   # :110-122
        shuffled, order_labels = create_random_shuffle(real_imgs)
        order_logits = netD.seq_consisten_model(shuffled)
        consistency, _ = F.MlsmFn.apply(order_logits, order_labels.unsqueeze(-1), 1)
        errD = _plus(errD, cfg.CONSISTENCY_RATIO, consistency)
        consistency = consistency.detach()
    return errD, parts[0], parts[1], parts[2], acc, consistency


def compute_generator_loss(netD, fake_imgs, real_imgs, real_labels, fake_catelabels, conditions, gpus):
    """reference miscc/utils.py:126-171. Returns (errD_fake, acc, consistency_loss_val)."""
    if conditions is None:
        raise NotImplementedError("unconditional critics are never built by trainer.py")
    cond = conditions.detach()
    fake_features = netD(fake_imgs)                                            # :137
    errD_fake = _bce(netD.get_cond_logits(fake_features, cond), real_labels)   # :139-141
    acc = 0
    if netD.cate_classify is not None:                                         # :149-153 fake feats vs REAL labels
        cate_loss, acc = _mlsm(netD.cate_classify(fake_features), fake_catelabels)
        errD_fake = errD_fake + cate_loss
    consistency = 0
    if netD.seq_consisten_model is not None:                                   # :155-169: real first, then fake
        real_logits = netD.seq_consisten_model(real_imgs)
        fake_logits = netD.seq_consisten_model(fake_imgs)
        consistency = F.MseFn.apply(fake_logits, real_logits.detach())
        errD_fake = _plus(errD_fake, cfg.CONSISTENCY_RATIO, consistency)
        consistency = consistency.detach()
    return errD_fake, acc, consistency


def KL_loss(mu, logvar):
    """-0.5 * mean(1 + logvar - mu^2 - exp(logvar))  (reference miscc/utils.py:184-188)."""
    return F.KlFn.apply(mu, logvar)


def mse_loss(a, b):
    """nn.MSELoss (reference trainer.py:222) on fp32 images or NCHW-shaped latent views."""
    if a.dim() == 4 and a.stride(1) == 1:        # latent views of NHWC storage: compare in storage order
        a, b = a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)
    return F.MseFn.apply(a, b)


_INIT_RULES = (          # substring of the class name -> (weight mean, weight std, zero the bias)
    ("Conv", 0.0, 0.02, False),
    ("BatchNorm", 1.0, 0.02, True),
    ("Linear", 0.0, 0.02, True),
)


def weights_init(m):
    """Initialiser passed to `net.apply()` (reference miscc/utils.py:191-201). Dispatch is on a substring of the class
    NAME, first match wins, which is what makes cpcsv.modules.Conv2d/HeadConv2d/BatchNorm*/Linear initialise like the
    torch.nn classes the reference builds; GRUCell matches nothing and keeps its default init. Conv biases are left
    alone, BatchNorm and Linear biases are zeroed."""
    kind = type(m).__name__
    for key, mean, std, zero_bias in _INIT_RULES:
        if key in kind:
            with torch.no_grad():
                m.weight.normal_(mean, std)
                if zero_bias and getattr(m, "bias", None) is not None:
                    m.bias.zero_()
            return


def get_multi_acc(predict, real):
    """reference miscc/utils.py:313-321 on numpy arrays (kept for API parity; the step uses
    multi_acc_device). Divides by zero like the reference if `real` has no positives."""
    predict = 1 / (1 + np.exp(-np.asarray(predict)))
    real = np.asarray(real)
    return float(np.sum((real == 1) & (predict >= 0.5))) / float(np.sum(real))


def save_model(netG, netD_im, netD_st, netD_se, epoch, model_dir, whole=False):
    """Checkpoint files with the reference's names (miscc/utils.py:323-338; read back by trainer.py:121-131 and
    inference.py:77-81): the generator is kept per epoch, the critics are overwritten. The wire format is the
    state_dict key set; `whole=True` pickles the modules instead (reference's `whole` branch)."""
    critics = (("netD_im", netD_im), ("netD_st", netD_st), ("netD_se", netD_se))
    if whole:
        for name, net in (("netG", netG),) + critics:
            if net is not None:
                torch.save(net, os.path.join(model_dir, name + ".pkl"))
        print('Save G/D model')
        return
    torch.save(netG.state_dict(), os.path.join(model_dir, "netG_epoch_%d.pth" % epoch))
    for name, net in critics:
        if net is not None:
            torch.save(net.state_dict(), os.path.join(model_dir, name + "_epoch_last.pth"))
    print('Save G/D models')


# ---------------------------------------------------------------------------------------------------------------
# F3: sample dumps (reference miscc/utils.py:204-311,343-364,402-428). CPU-side I/O around netG.sample_videos; the grid
# layout is torchvision.utils.make_grid's (0.4.2, requirements.txt:46), restated here because torchvision is not part
# of this image: nrow images per row, `padding` pixels of pad_value between and around them, one-channel images
# replicated to three, a single image returned unpadded.
# ---------------------------------------------------------------------------------------------------------------
def make_grid(tensor, nrow=8, padding=2, pad_value=0.0):
    if isinstance(tensor, (list, tuple)):
        tensor = torch.stack([t if t.dim() == 3 else t.unsqueeze(0) for t in tensor], 0)
    if tensor.dim() == 2:
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 3:
        if tensor.size(0) == 1:
            tensor = torch.cat((tensor, tensor, tensor), 0)
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 4 and tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(np.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, y * height + padding, height - padding).narrow(2, x * width + padding, width - padding).copy_(tensor[k])
            k += 1
    return grid


def images_to_numpy(tensor):
    """reference miscc/utils.py:229-234: (C,H,W) in [-1,1] -> (H,W,C) uint8."""
    generated = tensor.detach().float().cpu().numpy().transpose(1, 2, 0).copy()
    generated[generated < -1] = -1
    generated[generated > 1] = 1
    generated = (generated + 1) / 2 * 255
    return generated.astype('uint8')


def save_image(tensor, path, nrow=8, padding=2):
    """torchvision.utils.save_image as the reference calls it (no normalisation): [0,1] floats -> PNG."""
    import PIL.Image
    grid = make_grid(tensor.detach().float().cpu(), nrow=nrow, padding=padding)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    PIL.Image.fromarray(arr).save(path)


def save_story_results(ground_truth, images, texts, name, image_dir, step=0, lr=False):
    """reference miscc/utils.py:236-281: one row per story (its T frames side by side), generated | ground truth; writes
    the captions next to it and returns the uint8 sheet (the reference's PNG save is commented out there too)."""
    video_len = cfg.VIDEO_LEN
    sheet = lambda vids: images_to_numpy(make_grid([make_grid(torch.transpose(v, 0, 1), video_len) for v in vids.detach().float().cpu()], 1))
    all_images = sheet(images)
    if ground_truth is not None:
        all_images = np.concatenate([all_images, sheet(ground_truth)], axis=1)
    if texts is not None:
        with open('{}/fake_samples_{}.txt'.format(image_dir, name), 'w') as fid:
            for idx in range(images.shape[0]):
                fid.write(str(idx) + '--------------------------------------------------------\n')
                for i in range(len(texts)):
                    fid.write(texts[i][idx] + '\n')
                fid.write('\n\n')
    return all_images


def save_image_results(ground_truth, images, size=None):
    """reference miscc/utils.py:283-301: (ST*T, C, size, size) frames -> the same sheet layout."""
    video_len, st_bs = cfg.VIDEO_LEN, cfg.TRAIN.ST_BATCH_SIZE
    size = size or cfg.IMSIZE
    sheet = lambda x: images_to_numpy(make_grid([make_grid(v, video_len) for v in x.detach().float().cpu().reshape(st_bs, video_len, -1, size, size)], 1))
    all_images = sheet(images)
    if ground_truth is not None:
        all_images = np.concatenate([all_images, sheet(ground_truth)], axis=1)
    return all_images


def save_all_img(images, count, image_dir):
    """reference miscc/utils.py:303-311: every frame of (B,C,T,H,W) as <count>.png."""
    bs, _, v_len = images.shape[0], images.shape[1], images.shape[2]
    for b in range(bs):
        imgs = images[b].transpose(0, 1)
        for i in range(v_len):
            count += 1
            save_image(imgs[i], os.path.join(image_dir, "{}.png".format(count)))
    return count


def save_test_samples(netG, dataloader, save_path):
    """reference miscc/utils.py:343-371 (save_train_samples :373-400 differs only in the file-name width): run
    netG.sample_videos over a loader, dump caption sheets, images.npy and labels.npy."""
    print('Generating Test Samples...')
    dev = next(netG.parameters()).device
    save_images, save_labels = [], []
    for i, batch in enumerate(dataloader, 0):
        real_cpu = batch['images']
        text = batch['description'][:, :, :cfg.TEXT.DIMENSION].to(dev)
        catelabel = batch['labels'].to(dev)
        motion_input = torch.cat((text, catelabel), 2)
        with torch.no_grad():
            _, fake, _, _, _, _, _ = netG.sample_videos(motion_input, text)
        save_story_results(real_cpu, fake, batch.get('text'), '{:03d}'.format(i), save_path)
        save_images.append(fake.detach().cpu().numpy())
        save_labels.append(catelabel.detach().cpu().numpy())
    np.save(save_path + '/images.npy', np.concatenate(save_images, 0))
    np.save(save_path + '/labels.npy', np.concatenate(save_labels, 0))


save_train_samples = save_test_samples


def mkdir_p(path):
    os.makedirs(path, exist_ok=True)


def count_param(model):
    return sum(p.numel() for p in model.parameters())
