"""Networks of the CP-CSV story GAN, restated for CPU fp32. TEST INFRASTRUCTURE ONLY.

Restates (does not import) /root/reference/model.py, cascade_model.py and layers.py.
state_dict keys equal the reference's so reference checkpoints and
the golden fixtures load directly. Differences in *form* are deliberate: explicit config
object instead of a global, explicit noise source instead of hidden RNG draws, spectral
norm written out instead of the torch hook.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# spectral norm, written out  (torch.nn.utils.spectral_norm as used at model.py:19,79,
# 502-510,583-594; semantics in SURVEY Appendix B)
# --------------------------------------------------------------------------------------
class SpectralConv2d(nn.Module):
    """Conv2d whose weight is divided by its largest singular value estimate.

    One power iteration per forward call in train mode (also under no_grad), none in
    eval; u/v are buffers updated in place; gradient flows through sigma = u^T W v with
    u, v treated as constants.
    """

    def __init__(self, cin, cout, k, stride, pad, bias):
        super().__init__()
        self.stride, self.pad = stride, pad
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(cin * k * k)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight_u", F.normalize(torch.randn(cout), dim=0, eps=1e-12))
        self.register_buffer("weight_v", F.normalize(torch.randn(cin * k * k), dim=0, eps=1e-12))

    def normalised_weight(self):
        w = self.weight_orig
        wm = w.reshape(w.shape[0], -1)
        if self.training:
            with torch.no_grad():
                v = F.normalize(torch.mv(wm.t(), self.weight_u), dim=0, eps=1e-12)
                u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
                self.weight_v.copy_(v)
                self.weight_u.copy_(u)
        u, v = self.weight_u.clone(), self.weight_v.clone()
        sigma = torch.dot(u, torch.mv(wm, v))
        return w / sigma

    def forward(self, x):
        return F.conv2d(x, self.normalised_weight(), self.bias, self.stride, self.pad)


class SpectralWeight(nn.Module):
    """A weight of any rank divided by its largest singular value estimate: the same power iteration as
    SpectralConv2d on W.reshape(out, -1) (torch.nn.utils.spectral_norm as applied to the Conv3d / Linear layers of
    VideoEncoder, model.py:18-28,117-160). `op(x, w, b)` is the layer's functional form."""

    def __init__(self, shape, fan_in, bias, op):
        super().__init__()
        self.op = op
        w = torch.empty(*shape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(fan_in)
            self.bias = nn.Parameter(torch.empty(shape[0]).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight_u", F.normalize(torch.randn(shape[0]), dim=0, eps=1e-12))
        self.register_buffer("weight_v", F.normalize(torch.randn(w[0].numel()), dim=0, eps=1e-12))

    normalised_weight = SpectralConv2d.normalised_weight

    def forward(self, x):
        return self.op(x, self.normalised_weight(), self.bias)


def _sn_conv3d(cin, cout, k, stride, pad):
    return SpectralWeight((cout, cin) + tuple(k), cin * k[0] * k[1] * k[2], False,
                          lambda x, w, b: F.conv3d(x, w, b, stride, pad))


def _sn_linear(cin, cout):
    return SpectralWeight((cout, cin), cin, True, F.linear)


class OrderCritic(nn.Module):
    """VideoEncoder, model.py:99-210: the optional order-consistency critic on whole stories (B,3,T,64,64) -> (B,1)
    logit. A (2+1)D tower: spatial (1,k,k) stride-(1,2,2) convs alternate with temporal (3,1,1) stride-(2,1,1) convs, every
    conv spectral-normed and followed by BatchNorm3d; the 1x1x1 conv of the stem pads TIME by 1 on both sides
    (model.py:24-26), so T grows from 5 to 7 there; global average pool; SN-Linear(512,128)+BN1d+ReLU+SN-Linear(128,1).
    Module indices equal the reference's (state_dict keys: story_encoder.0.{0,1,3,4}.*, story_encoder.{1..24}.*,
    detector.{0,1,3}.*)."""

    def __init__(self):
        super().__init__()
        stem = nn.Sequential(_sn_conv3d(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)), nn.BatchNorm3d(45), nn.ReLU(),
                             _sn_conv3d(45, 64, (1, 1, 1), (1, 1, 1), (1, 0, 0)), nn.BatchNorm3d(64), nn.ReLU())
        block = [stem]
        spatial = lambda ci, co: _sn_conv3d(ci, co, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        temporal = lambda ci, co: _sn_conv3d(ci, co, (3, 1, 1), (2, 1, 1), (1, 0, 0))
        for conv, ci, co in ((spatial, 64, 128), (temporal, 128, 128), (spatial, 128, 128), (temporal, 128, 256),
                             (spatial, 256, 256), (temporal, 256, 512), (spatial, 512, 512), (temporal, 512, 512)):
            block += [conv(ci, co), nn.BatchNorm3d(co), nn.LeakyReLU(0.2)]
        self.pool = nn.AdaptiveAvgPool3d(1)
        self.story_encoder = nn.Sequential(*block)
        self.detector = nn.Sequential(_sn_linear(512, 128), nn.BatchNorm1d(128), nn.ReLU(), _sn_linear(128, 1))

    def forward(self, story):
        b = story.shape[0]
        return self.detector(self.pool(self.story_encoder(story)).view(b, -1))


def order_critic_state(seed):
    """A reproducible state_dict for OrderCritic / the reference's VideoEncoder (4.6 M parameters - too large to commit
    as a fixture): weights_init statistics (conv / linear weights N(0,.02), BatchNorm gains N(1,.02), biases 0),
    unit-norm random u/v, fresh running statistics, all drawn from ONE seeded CPU generator in state_dict order."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderCritic().state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_orig"):
            out[k] = torch.randn(v.shape, generator=g) * 0.02
        elif k.endswith(("weight_u", "weight_v")):
            out[k] = F.normalize(torch.randn(v.shape, generator=g), dim=0, eps=1e-12)
        elif k.endswith("running_mean") or k.endswith(".bias"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = torch.ones_like(v)
        elif k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith(".weight"):                      # BatchNorm gains
            out[k] = 1.0 + torch.randn(v.shape, generator=g) * 0.02
        else:
            raise KeyError(k)
    return out


def _conv3x3(cin, cout):
    """model.py:16-22 — 3x3, stride 1, pad 1, no bias."""
    return nn.Conv2d(cin, cout, 3, 1, 1, bias=False)


def _grow(cin, cout):
    """upBlock, model.py:26-34: nearest x2 -> conv3x3 -> BatchNorm2d -> ReLU."""
    return nn.Sequential(nn.Upsample(scale_factor=2, mode="nearest"), _conv3x3(cin, cout),
                         nn.BatchNorm2d(cout), nn.ReLU())


def _shrink(cin, cout):
    """downBlock, cascade_model.py:36-41: conv3x3 stride 2 WITH bias -> BN -> ReLU."""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 2, 1, bias=True), nn.BatchNorm2d(cout), nn.ReLU())


class CondAugment(nn.Module):
    """CA_NET, model.py:37-65. ReLU is applied BEFORE the mu/logvar split (quirk A4)."""

    def __init__(self, cfg):
        super().__init__()
        self.c = cfg.cond_dim
        self.fc = nn.Linear(cfg.text_dim * cfg.video_len, 2 * cfg.cond_dim)

    def forward(self, text, noise):
        h = torch.relu(self.fc(text))
        mu, logvar = h[:, :self.c], h[:, self.c:]
        eps = noise((mu.shape[0], self.c))            # model.py:56-58
        return eps * torch.exp(0.5 * logvar) + mu, mu, logvar


def dynamic_filter_1d(signal, taps, pad):
    """DynamicFilterLayer1D.forward, layers.py:69-80, as ONE batched op.

    signal (N,C,L), taps (N,1,C,K) -> (N,1,L): per-sample cross-correlation, zero pad.
    The reference loops N conv1d calls and concatenates; a grouped conv is the same sum.
    """
    n, c, length = signal.shape
    k = taps.shape[-1]
    out = F.conv1d(signal.reshape(1, n * c, length), taps.reshape(n, c, k), padding=pad, groups=n)
    return out.reshape(n, 1, length)


class StoryGenerator(nn.Module):
    """StoryGAN (plain), model.py:214-483."""

    cascade = False

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        md, cd, nz = cfg.motion_dim, cfg.cond_dim, cfg.z_dim
        ngf, nseg = cfg.ngf, cfg.gf_seg_dim
        self.recurrent = nn.GRUCell(nz + md, md)                      # model.py:223
        self.mocornn = nn.GRUCell(md, cd)                             # model.py:224
        self.ca_net = CondAugment(cfg)
        nfilt = cfg.dfl_taps * cfg.dfl_channels
        self.filter_net = nn.Sequential(nn.Linear(cd, nfilt), nn.BatchNorm1d(nfilt))      # :250-252
        nimg = cfg.dfl_width * cfg.dfl_channels
        self.image_net = nn.Sequential(nn.Linear(md, nimg), nn.BatchNorm1d(nimg), nn.Tanh())  # :254-257
        self.fc = nn.Sequential(nn.Linear(cfg.joint_dim, ngf * 16, bias=False),
                                nn.BatchNorm1d(ngf * 16), nn.ReLU())                      # :260-263
        for i in range(4):                                                                 # :264-270
            setattr(self, "upsample%d" % (i + 1), _grow(ngf >> i, ngf >> (i + 1)))
        self.img = nn.Sequential(_conv3x3(ngf // 16, 3), nn.Tanh())                       # :272-274
        if cfg.segment_learning:
            self.seg_c = _conv3x3(nseg, ngf)                                              # :278
            self.seg_c1 = _conv3x3(nseg // 2, ngf // 2)                                   # :279
            self.fc_seg = nn.Sequential(nn.Linear(cfg.joint_dim, nseg * 16, bias=False),
                                        nn.BatchNorm1d(nseg * 16), nn.ReLU())             # :285-288
            for i in range(4):                                                             # :290-296
                setattr(self, "upsample%d_seg" % (i + 1), _grow(nseg >> i, nseg >> (i + 1)))
            self.img_seg = nn.Sequential(_conv3x3(nseg // 16, 1), nn.Tanh())              # :298-300
            self._extra_seg_modules(nseg)
        self.m_net = nn.Sequential(nn.Linear(md, md), nn.BatchNorm1d(md))                 # :302-304
        self.c_net = nn.Sequential(nn.Linear(cd, cd), nn.BatchNorm1d(cd))                 # :306-308

    def _extra_seg_modules(self, nseg):
        pass

    # -- recurrent text encoders ------------------------------------------------------
    def _motion_codes(self, motion, steps, noise):
        """sample_z_motion, model.py:321-334. motion (B,T,md) or (B,md)."""
        b = motion.shape[0]
        h = self.m_net(noise((b, self.cfg.motion_dim)))                # :319,324
        outs = []
        for t in range(steps):
            m_t = motion if motion.dim() == 2 else motion[:, t]
            z = noise((b, self.cfg.z_dim))                             # :315
            h = self.recurrent(torch.cat((z, m_t), 1), h)
            outs.append(h)
        return torch.stack(outs, 1).reshape(-1, self.cfg.motion_dim)   # story-major rows

    def _content_codes(self, motion, content):
        """motion_content_rnn, model.py:336-346."""
        h = self.c_net(content)
        if motion.dim() == 2:
            motion = motion.unsqueeze(1)
            steps = 1
        else:
            steps = self.cfg.video_len
        outs = []
        for t in range(steps):
            h = self.mocornn(motion[:, t], h)
            outs.append(h)
        return torch.stack(outs, 1).reshape(-1, self.cfg.cond_dim)

    def _joint_code(self, frame_motion, zm, c_rows, crnn):
        """model.py:371-378 / :436-443 — concat + dynamic filter."""
        cfg = self.cfg
        zmc = torch.cat((zm, c_rows), 1)
        sig = self.image_net(frame_motion).view(-1, cfg.dfl_channels, cfg.dfl_width)
        taps = self.filter_net(crnn).view(-1, 1, cfg.dfl_channels, cfg.dfl_taps)
        mixed = dynamic_filter_1d(sig, taps, cfg.dfl_taps // 2)
        return torch.cat((zmc, mixed.squeeze(1)), 1)

    # -- image decoder ----------------------------------------------------------------
    def _decode(self, joint):
        """model.py:379-405 (plain). Returns (latents, rgb, seg)."""
        cfg = self.cfg
        x = self.fc(joint).view(-1, cfg.ngf, 4, 4)
        if not cfg.segment_learning:
            for i in range(4):
                x = getattr(self, "upsample%d" % (i + 1))(x)
            return None, self.img(x), None
        s = self.fc_seg(joint).view(-1, cfg.gf_seg_dim, 4, 4)
        x = self.seg_c(s) * x + x                                       # :383
        s = self.upsample1_seg(s)
        x = self.upsample1(x)
        x = self.seg_c1(s) * x + x                                      # :387
        for i in (2, 3, 4):
            s = getattr(self, "upsample%d_seg" % i)(s)
            x = getattr(self, "upsample%d" % i)(x)
        return None, self.img(x), self.img_seg(s)

    # -- public surface (same tuples as the reference) ----------------------------------
    def sample_videos(self, motion, content, seg=False, noise=None):
        """model.py:348-423. motion (B,T,md), content (B,T,text) -> 7-tuple."""
        noise = noise or _default_noise
        cfg = self.cfg
        b, t = motion.shape[0], motion.shape[1]
        text = content.reshape(-1, cfg.video_len * content.shape[2])
        r_code, r_mu, r_logvar = self.ca_net(text, noise)
        c_rows = r_mu.repeat(cfg.video_len, 1)                          # :361 tiled, NOT story-major
        crnn = self._content_codes(motion, r_code)                      # :364 sampled code
        flat = motion.reshape(-1, motion.shape[2])
        zm = self._motion_codes(motion, cfg.video_len, noise)           # :368
        joint = self._joint_code(flat, zm, c_rows, crnn)
        lat, rgb, segm = self._decode(joint)
        video = rgb.view(b, t, 3, rgb.shape[-2], rgb.shape[-1]).permute(0, 2, 1, 3, 4)  # :406-407
        return lat, video, flat, flat, r_mu, r_logvar, (segm if seg else None)

    def sample_images(self, motion, content, seg=False, noise=None):
        """model.py:426-483. motion (B,md), content (B,T,text)."""
        noise = noise or _default_noise
        cfg = self.cfg
        text = content.reshape(-1, cfg.video_len * content.shape[2])
        _, c_mu, c_logvar = self.ca_net(text, noise)
        crnn = self._content_codes(motion, c_mu)                        # :433 the MEAN, quirk 2
        zm = self._motion_codes(motion, 1, noise)
        joint = self._joint_code(motion, zm, c_mu, crnn)
        lat, rgb, segm = self._decode(joint)
        return lat, rgb, motion, motion, c_mu, c_logvar, (segm if seg else None)


class CascadeStoryGenerator(StoryGenerator):
    """StoryGAN (cascade), cascade_model.py:221-540: the generated segmentation image is
    re-encoded and THOSE features gate the image branch."""

    cascade = True

    def _extra_seg_modules(self, nseg):
        self.presample = nn.Sequential(_conv3x3(1, nseg // 16), nn.BatchNorm2d(nseg // 16), nn.ReLU())  # :312-316
        for i in range(4):                                                                               # :317-320
            setattr(self, "downsample%d_seg" % (i + 1), _shrink(nseg >> (4 - i), nseg >> (3 - i)))

    def _encode_seg(self, seg_img):
        g = [self.presample(seg_img)]
        for i in range(4):
            g.append(getattr(self, "downsample%d_seg" % (i + 1))(g[-1]))
        return g  # [latent64, g_seg4 (32), g_seg3 (16), g_seg2 (8), g_seg1 (4)]

    def _decode(self, joint):
        """cascade_model.py:401-438."""
        cfg = self.cfg
        x = self.fc(joint).view(-1, cfg.ngf, 4, 4)
        s0 = self.fc_seg(joint).view(-1, cfg.gf_seg_dim, 4, 4)
        h = [s0]
        for i in range(4):
            h.append(getattr(self, "upsample%d_seg" % (i + 1))(h[-1]))
        segm = self.img_seg(h[4])
        _, g4, g3, g2, g1 = self._encode_seg(segm)
        x = self.seg_c(g1) * x + x                                      # :419
        x = self.upsample1(x)
        x = self.seg_c1(g2) * x + x                                     # :423
        for i in (2, 3, 4):
            x = getattr(self, "upsample%d" % i)(x)
        return ((h[0], h[1], h[2], h[3]), (g1, g2, g3, g4)), self.img(x), segm   # :441

    def train_autoencoder(self, seg_img):
        """cascade_model.py:528-540."""
        g = self._encode_seg(seg_img)
        h = g[4]
        for i in range(4):
            h = getattr(self, "upsample%d_seg" % (i + 1))(h)
        return self.img_seg(h)


def _default_noise(shape):
    return torch.empty(shape).normal_()


# --------------------------------------------------------------------------------------
# critics
# --------------------------------------------------------------------------------------
class CondLogits(nn.Module):
    """D_GET_LOGITS (bcondition=True), model.py:68-97."""

    def __init__(self, ndf, nef):
        super().__init__()
        self.nef = nef
        self.outlogits = nn.Sequential(
            SpectralConv2d(ndf * 8 + nef, ndf * 8, 3, 1, 1, bias=False),   # :76
            nn.BatchNorm2d(ndf * 8),
            nn.LeakyReLU(0.2),
            SpectralConv2d(ndf * 8, 1, 4, 4, 0, bias=True),                # :79
            nn.Sigmoid())

    def forward(self, feat, cond=None):
        if cond is not None:
            tiled = cond.view(-1, self.nef, 1, 1).repeat(1, 1, 4, 4)      # :89-90
            feat = torch.cat((feat, tiled), 1)
        return self.outlogits(feat).view(-1)


def _tower(cin, ndf, first_spectral):
    """encode_img, model.py:498-514 / 540-556 / 582-598: four conv4x4 s2 p1 stages."""
    first = SpectralConv2d(cin, ndf, 4, 2, 1, bias=False) if first_spectral \
        else nn.Conv2d(cin, ndf, 4, 2, 1, bias=False)
    layers = [first, nn.LeakyReLU(0.2)]
    c = ndf
    for _ in range(3):
        layers += [SpectralConv2d(c, 2 * c, 4, 2, 1, bias=False), nn.BatchNorm2d(2 * c), nn.LeakyReLU(0.2)]
        c *= 2
    return nn.Sequential(*layers)


class _CriticBase(nn.Module):
    in_channels = 3
    first_spectral = False
    has_classifier = True
    has_order_critic = False

    def __init__(self, cfg):
        super().__init__()
        ndf = cfg.df_dim
        self.encode_img = _tower(self.in_channels, ndf, self.first_spectral)
        self.seq_consisten_model = OrderCritic() if (self.has_order_critic and cfg.use_seq_consistency) else None   # model.py:599-601
        self.get_cond_logits = CondLogits(ndf, cfg.critic_cond_dim)        # model.py:516
        self.get_uncond_logits = None
        self.cate_classify = nn.Conv2d(ndf * 8, cfg.label_num, 4, 4, 1, bias=False) \
            if self.has_classifier else None                                 # model.py:520

    def forward(self, x):
        return self.encode_img(x)


class FrameCritic(_CriticBase):
    """STAGE1_D_IMG, model.py:487-527."""


class SegCritic(_CriticBase):
    """STAGE1_D_SEG, model.py:529-569."""
    in_channels = 1


class StoryCritic(_CriticBase):
    """STAGE1_D_STY_V2, model.py:571-618: frames folded into batch, features averaged over T."""
    first_spectral = True
    has_classifier = False
    has_order_critic = True

    def forward(self, story):
        n, c, t, hh, ww = story.shape
        frames = story.permute(0, 2, 1, 3, 4).contiguous().view(-1, c, hh, ww)   # :612-613
        f = torch.squeeze(self.encode_img(frames))                                 # :614
        f = f.view(n, t, *f.shape[1:])
        return f.mean(1).squeeze()                                                 # :617


def init_like_reference(net):
    """weights_init, miscc/utils.py:191-201, applied via .apply() (children first).

    Conv* -> N(0,.02) (bias untouched), BatchNorm* -> w N(1,.02), b 0, Linear -> N(0,.02), b 0.
    GRUCell keeps its default U(+-1/sqrt(H)) (SURVEY A15 [probe]). Spectral convs receive
    the same N(0,.02) on weight_orig (the reference's .weight shares storage before the
    first forward)."""
    def visit(m):
        if isinstance(m, (nn.Conv2d,)):
            m.weight.data.normal_(0.0, 0.02)
        elif isinstance(m, SpectralConv2d):
            m.weight_orig.data.normal_(0.0, 0.02)
        elif isinstance(m, SpectralWeight):
            # class-name dispatch of the reference: 'Conv3d' hits the Conv branch (weight only), 'Linear' the Linear
            # branch (weight N(0,.02), bias 0)
            m.weight_orig.data.normal_(0.0, 0.02)
            if m.weight_orig.dim() == 2 and m.bias is not None:
                m.bias.data.fill_(0.0)
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.weight.data.normal_(1.0, 0.02)
            m.bias.data.fill_(0)
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.fill_(0.0)
    net.apply(visit)
    return net
