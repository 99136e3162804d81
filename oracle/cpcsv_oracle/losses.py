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


def critic_loss(net, real, fake, ones, zeros, labels, cond):
    """compute_discriminator_loss, miscc/utils.py:48-123 (conditional branch, no uncond head,
    no sequence-consistency model). Returns the reference's 6-tuple."""
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
    return total, e_real.detach(), e_wrong.detach(), e_fake.detach(), acc, 0


def generator_loss(net, fake, real, ones, labels, cond):
    """compute_generator_loss, miscc/utils.py:126-171. The class loss uses FAKE features
    against the REAL labels (quirk 6)."""
    cond = cond.detach()
    f_fake = net(fake)                                                   # :137
    err = F.binary_cross_entropy(net.get_cond_logits(f_fake, cond), ones)             # :139-141
    acc = 0
    if net.cate_classify is not None:                                                  # :149-153
        cl = net.cate_classify(f_fake).squeeze()
        err = err + 1.0 * F.multilabel_soft_margin_loss(cl, labels)
        acc = multilabel_hit_rate(cl.detach(), labels)
    return err, acc, 0
