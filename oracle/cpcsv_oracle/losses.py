"""Loss glue of the CP-CSV step, restated. TEST INFRASTRUCTURE ONLY.

Restates /root/reference/miscc/utils.py:48-201,313-321 (no data_parallel: one device).
"""
import torch
import torch.nn.functional as F


def multilabel_hit_rate(logits, labels):
    """get_multi_acc, miscc/utils.py:313-321: #(label==1 and sigmoid(logit)>=.5) / #(label==1)."""
    hit = ((labels == 1) & (torch.sigmoid(logits) >= 0.5)).sum().item()
    return hit / float(labels.sum().item())


def kl_term(mu, logvar):
    """KL_loss, miscc/utils.py:184-188: -0.5 * mean(1 + logvar - mu^2 - exp(logvar))."""
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


def shuffle_plan(n_stories, video_len, np_rng, py_rng, random_rate=0.5):
    """The decisions of create_random_shuffle, miscc/utils.py:17-44, in the reference's draw order (numpy for the
    coin flips and re-shuffles, python `random` for the permutation, the donor story and the mixed-in slot).
    Returns (labels[n], src_story[n][T], src_frame[n][T]): output frame t of story b = input frame src_frame[b][t] of
    story src_story[b][t]. label 1 = shuffled (never left sorted) and, unless the donor is the story itself, ONE slot
    overwritten by the donor story's frame at that same slot."""
    import numpy as np
    labels, src_story, src_frame = [], [], []
    for idx in range(n_stories):
        label = 1 if random_rate > np_rng.random() else 0                     # :25
        ss, sf = [idx] * video_len, list(range(video_len))
        if label == 1:
            seq = py_rng.sample(range(video_len), video_len)                   # :29
            while bool((np.diff(seq) >= 0).all()):                             # :30 make sure not sorted
                np_rng.shuffle(seq)
            sf = list(seq)
            donor = py_rng.randint(0, n_stories - 1)                           # :33
            if donor != idx:
                slot = py_rng.sample(range(video_len), 1)[0]                   # :35
                ss[slot], sf[slot] = donor, slot                               # :36 stories[donor, :, slot]
        labels.append(label)
        src_story.append(ss)
        src_frame.append(sf)
    return labels, src_story, src_frame


def apply_shuffle(stories, plan):
    """stories (B,C,T,H,W) -> shuffled copy, order labels (B,) float."""
    labels, ss, sf = plan
    b, c, t = stories.shape[:3]
    out = torch.stack([torch.stack([stories[ss[i][k], :, sf[i][k]] for k in range(t)], 1) for i in range(b)], 0)
    return out, torch.tensor(labels, dtype=torch.float32)


def critic_loss(net, real, fake, ones, zeros, labels, cond, shuffle=None, consistency_ratio=1.0):
    """compute_discriminator_loss, miscc/utils.py:48-123 (conditional branch, no uncond head). With an order critic
    (`net.seq_consisten_model`, :110-122) `shuffle` is the shuffle_plan for the real stories. Returns the reference's
    6-tuple."""
    n = real.size(0)
    cond = cond.detach()
    f_real = net(real)                                                   # :70
    f_fake = net(fake.detach())                                          # :54,71
    e_real = F.binary_cross_entropy(net.get_cond_logits(f_real, cond), ones)          # :74-76
    e_wrong = F.binary_cross_entropy(net.get_cond_logits(f_real[:n - 1], cond[1:]), zeros[1:])  # :78-80
    e_fake = F.binary_cross_entropy(net.get_cond_logits(f_fake, cond), zeros)         # :82-84
    total = e_real + (e_fake + e_wrong) * 0.5                                          # :101
    acc = 0
    if net.cate_classify is not None:                                                  # :104-108
        cl = net.cate_classify(f_real).squeeze()
        total = total + 1.0 * F.multilabel_soft_margin_loss(cl, labels)
        acc = multilabel_hit_rate(cl.detach(), labels)
    cons = 0
    if getattr(net, "seq_consisten_model", None) is not None:                          # :110-122
        shuffled, order = apply_shuffle(real, shuffle)
        logits = net.seq_consisten_model(shuffled)
        c = F.binary_cross_entropy_with_logits(logits, order.to(logits.dtype).unsqueeze(-1))
        total = total + consistency_ratio * c
        cons = c.item()
    return total, e_real.detach(), e_wrong.detach(), e_fake.detach(), acc, cons


def generator_loss(net, fake, real, ones, labels, cond, consistency_ratio=1.0):
    """compute_generator_loss, miscc/utils.py:126-171. The class loss uses FAKE features
    against the REAL labels (quirk 6). With an order critic (:155-169): MSE between its logit on the fake stories and
    its (detached) logit on the real ones - real first, then fake, both in train mode."""
    cond = cond.detach()
    f_fake = net(fake)                                                   # :137
    err = F.binary_cross_entropy(net.get_cond_logits(f_fake, cond), ones)             # :139-141
    acc = 0
    if net.cate_classify is not None:                                                  # :149-153
        cl = net.cate_classify(f_fake).squeeze()
        err = err + 1.0 * F.multilabel_soft_margin_loss(cl, labels)
        acc = multilabel_hit_rate(cl.detach(), labels)
    cons = 0
    if getattr(net, "seq_consisten_model", None) is not None:
        real_logits = net.seq_consisten_model(real)                                    # :165
        fake_logits = net.seq_consisten_model(fake)                                    # :166
        c = F.mse_loss(fake_logits, real_logits.detach())
        err = err + consistency_ratio * c
        cons = c.item()
    return err, acc, cons
