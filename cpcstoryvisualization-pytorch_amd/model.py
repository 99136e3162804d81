"""model.py — MI355X-native networks of the CP-CSV story GAN (plain generator + three critics).

Drop-in for the reference's model.py: same class names, constructor arguments, method names,
return tuples and state_dict keys (reference: /root/reference/model.py; SURVEY.md §8(a) A1-A12,
§8(f) F2). Nothing here calls torch.nn compute: every layer executes the HIP kernels of
libcpcsv_hip.so through cpcsv.modules / cpcsv.functional. `STAGE1_G` is an alias of `StoryGAN`
(the name BASELINE.json's north_star uses; the reference only has it as a config key).

Differences that are deliberate and documented in DESIGN.md:
  * noise can be injected (`netG.noise_source = callable(shape)->tensor`) so parity tests replay
    the reference's draws; by default it is drawn on the device in the reference's order.
"""
import contextlib
import os

import torch
import torch.nn as nn

from cpcsv import functional as F
from cpcsv import modules as M
from cpcsv import textpath as TP
from cpcsv.runtime import branch, dcode, row_groups, tdtype
from miscc.config import cfg

# segmentation decoder of the NO-GRAD pass on its own stream (CPCSV_DEC_BRANCH=0: never). The same fork in the differentiable pass,
# forward or backward only, measured +0.2 ... +0.3 ms per step in round 4 (that pass shares the GPU with three critic updates) and
# is gone.
_DEC_BRANCH = os.environ.get("CPCSV_DEC_BRANCH", "1") != "0"
_NOISE_BANK = os.environ.get("CPCSV_NOISE_BANK", "1") != "0"
_PREPACK_TEXT = os.environ.get("CPCSV_PREPACK_TEXT", "1") != "0"
_TEXT_MODE = os.environ.get("CPCSV_TEXT_STREAMS", "1")
_TEXT_STREAMS = _TEXT_MODE != "0" and os.environ.get("CPCSV_STREAMS", "1") != "0"


def conv3x3(in_planes, out_planes, stride=1, use_spectral_norm=False):
    "3x3 convolution with padding, no bias (reference model.py:16-22)"
    return M.Conv2d(in_planes, out_planes, 3, stride, 1, bias=False, spectral=use_spectral_norm)


def upBlock(in_planes, out_planes):
    """nearest x2 -> conv3x3 -> BatchNorm2d -> ReLU (reference model.py:26-34) as ONE fused node:
    the upsample is an index shift in the conv's gather, BN statistics come out of the GEMM epilogue."""
    return M.FusedSequential(M.Upsample(), conv3x3(in_planes, out_planes), M.BatchNorm2d(out_planes), nn.ReLU(True))


class CA_NET(nn.Module):
    """Conditioning augmentation (reference model.py:37-65). ReLU precedes the mu/logvar split."""

    def __init__(self):
        super().__init__()
        self.t_dim = cfg.TEXT.DIMENSION * cfg.VIDEO_LEN
        self.c_dim = cfg.GAN.CONDITION_DIM
        self.fc = M.Linear(self.t_dim, self.c_dim * 2, bias=True)
        self.relu = nn.ReLU()
        self.noise_source = None

    def encode(self, text_embedding):
        lay = M._layer_for(self.fc, None, M.L.ACT_RELU, 0, out_mode="f32pad")
        x = lay(text_embedding)
        mu = F.UnpadFn.apply(x, 0, self.c_dim)
        logvar = F.UnpadFn.apply(x, self.c_dim, self.c_dim)
        return mu, logvar

    def reparametrize(self, mu, logvar):
        eps = _draw(self.noise_source, tuple(mu.shape), mu.device)       # model.py:56-58
        return F.ReparamFn.apply(mu, logvar, eps)

    def forward(self, text_embedding):
        mu, logvar = self.encode(text_embedding)
        return self.reparametrize(mu, logvar), mu, logvar


class _NoiseBank:
    """All N(0,1) draws of one generator pass as ONE launch: the reference draws CA eps, the GRU's initial-state noise and one
    noise vector per time step separately (model.py:56-58,315,319) - 14 tiny RNG launches per pass, at the head of four
    launch-latency-bound encoder chains. The bank is drawn before the chains fork and hands out consecutive views in call
    order; i.i.d. standard normals either way. Only when no noise source is injected (parity tests replay recorded draws)."""

    def __init__(self, total, device):
        self.buf = torch.randn(total, device=device, dtype=torch.float32)
        self.pos = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        if self.pos + n > self.buf.numel():
            return None
        v = self.buf[self.pos:self.pos + n].view(shape)
        self.pos += n
        return v


_BANK = [None]


def _draw(source, shape, device):
    if source is not None:
        return source(shape).to(device=device, dtype=torch.float32)
    if _BANK[0] is not None:
        v = _BANK[0].take(shape)
        if v is not None:
            return v
    return torch.randn(shape, device=device, dtype=torch.float32)


class D_GET_LOGITS(nn.Module):
    """Conditional logits head (reference model.py:68-97)."""

    def __init__(self, ndf, nef, bcondition=True):
        super().__init__()
        self.df_dim, self.ef_dim, self.bcondition = ndf, nef, bcondition
        if bcondition:
            self.outlogits = M.FusedSequential(
                conv3x3(ndf * 8 + nef, ndf * 8, use_spectral_norm=True),
                M.BatchNorm2d(ndf * 8),
                nn.LeakyReLU(0.2, inplace=True),
                M.Conv2d(ndf * 8, 1, 4, 4, 0, bias=True, spectral=True),
                nn.Sigmoid(), head_last=True)
            if os.environ.get("CPCSV_HEAD_DGRAD_COLS", "1") != "0":
                # the condition channels of the head's input get no gradient (detached by every caller): dX only for the features
                self.outlogits._plan()[0].dgrad_cols = ndf * 8
        else:
            self.outlogits = M.FusedSequential(M.Conv2d(ndf * 8, 1, 4, 4, 0, bias=True, spectral=True),
                                               nn.Sigmoid(), head_last=True)

    def forward(self, h_code, c_code=None):
        h = _as_nhwc(h_code)
        if self.bcondition and c_code is not None:
            cond = c_code.reshape(-1, self.ef_dim)
            plan = self.outlogits._plan()
            if plan[0].cond_head_ok(h, cond, None):
                # the condition channels are spatially constant: factored conv (csrc/condhead.hip), no concatenated tensor
                return plan[1](plan[0](h, cond=cond.detach())).view(-1)
            h = F.CondConcatFn.apply(h, cond, self.df_dim * 8)   # model.py:89-92
            h._cpcsv_live_cols = self.df_dim * 8           # (its backward reads only the feature columns of dX)
        return self.outlogits(h).view(-1)

    def forward_triplet(self, feats, c_code):
        """The three calls of a critic update (reference miscc/utils.py:74-84: real / wrong / fake) in ONE pass:
        feats = [real | fake] features (2N rows) -> probabilities [real (N) | wrong (N-1) | fake (N)]. Each of the three
        is its own BatchNorm batch and spectral-norm iteration (cpcsv.runtime.row_groups), in the reference's order."""
        h = _as_nhwc(feats)
        n = h.shape[0] // 2
        cond = c_code.reshape(-1, self.ef_dim)
        plan = self.outlogits._plan()
        if plan[0].cond_head_ok(h, cond, (n, n - 1, n)):
            # the three calls from 2 N distinct feature maps and N distinct condition rows (csrc/condhead.hip): no triplet tensor
            with row_groups((n, n - 1, n)):
                return plan[1](plan[0](h, cond=cond.detach())).view(-1)
        x = F.CondTripletFn.apply(h, cond, self.df_dim * 8)
        x._cpcsv_live_cols = self.df_dim * 8               # (its backward reads only the feature columns of dX)
        with row_groups((n, n - 1, n)):
            return self.outlogits(x).view(-1)


def _as_nhwc(feat):
    """Critic features travel as an NCHW-SHAPED view of NHWC storage (zero-copy both ways)."""
    if feat.dim() == 4 and feat.stride(1) == 1:
        return feat.permute(0, 2, 3, 1)
    return F.ToNhwcFn.apply(feat, tdtype())


class StoryGAN(nn.Module):
    """Text -> story generator (reference model.py:214-483)."""

    def __init__(self, video_len):
        super().__init__()
        self.gf_dim = cfg.GAN.GF_DIM * 8
        self.gf_dim_seg = cfg.GAN.GF_SEG_DIM
        self.motion_dim = cfg.TEXT.DIMENSION + cfg.LABEL_NUM
        self.content_dim = cfg.GAN.CONDITION_DIM
        self.noise_dim = cfg.GAN.Z_DIM
        self.recurrent = M.GRUCell(self.noise_dim + self.motion_dim, self.motion_dim)
        self.mocornn = M.GRUCell(self.motion_dim, self.content_dim)
        self.video_len = video_len
        self.n_channels = 3
        self.filter_num = 3
        self.filter_size = 21
        self.image_size = 124
        self.out_num = 1
        self.use_segment = cfg.SEGMENT_LEARNING
        self.segment_size = 4 * 2 ** 4                     # 4x4 seed map, four x2 up-blocks
        self.noise_source = None
        self.define_module()

    # -- construction (reference model.py:242-311) ------------------------------------------------
    def define_module(self):
        from layers import DynamicFilterLayer1D as DynamicFilterLayer
        ninput = self.motion_dim + self.content_dim + self.image_size
        ngf = self.gf_dim
        self.ca_net = CA_NET()
        nfilt = self.filter_size * self.filter_num * self.out_num
        self.filter_net = M.FusedSequential(M.Linear(self.content_dim, nfilt), M.BatchNorm1d(nfilt), out_mode="f32")
        nimg = self.image_size * self.filter_num
        self.image_net = M.FusedSequential(M.Linear(self.motion_dim, nimg), M.BatchNorm1d(nimg), nn.Tanh(), out_mode="f32")
        self.fc = M.FusedSequential(M.Linear(ninput, ngf * 4 * 4, bias=False), M.BatchNorm1d(ngf * 4 * 4), nn.ReLU(True),
                                    out_mode="T")
        self.upsample1 = upBlock(ngf, ngf // 2)
        self.upsample2 = upBlock(ngf // 2, ngf // 4)
        self.upsample3 = upBlock(ngf // 4, ngf // 8)
        self.upsample4 = upBlock(ngf // 8, ngf // 16)
        self.img = M.FusedSequential(conv3x3(ngf // 16, 3), nn.Tanh())
        if self.use_segment:
            ngf_seg = self.gf_dim_seg
            self.seg_c = conv3x3(ngf_seg, ngf)
            self.seg_c1 = conv3x3(ngf_seg // 2, ngf // 2)
            self.fc_seg = M.FusedSequential(M.Linear(ninput, ngf_seg * 4 * 4, bias=False), M.BatchNorm1d(ngf_seg * 4 * 4),
                                            nn.ReLU(True), out_mode="T")
            self.upsample1_seg = upBlock(ngf_seg, ngf_seg // 2)
            self.upsample2_seg = upBlock(ngf_seg // 2, ngf_seg // 4)
            self.upsample3_seg = upBlock(ngf_seg // 4, ngf_seg // 8)
            self.upsample4_seg = upBlock(ngf_seg // 8, ngf_seg // 16)
            self.img_seg = M.FusedSequential(conv3x3(ngf_seg // 16, 1), nn.Tanh())
            self._define_cascade(ngf_seg)
        # (padded fp32 outputs: they are the GRU states' initial values, which stay in the padded layout W_hh reads)
        self.m_net = M.FusedSequential(M.Linear(self.motion_dim, self.motion_dim), M.BatchNorm1d(self.motion_dim),
                                       out_mode="f32pad")
        self.c_net = M.FusedSequential(M.Linear(self.content_dim, self.content_dim), M.BatchNorm1d(self.content_dim),
                                       out_mode="f32pad")
        self.dfn_layer = DynamicFilterLayer(self.filter_size, pad=self.filter_size // 2)

    def _define_cascade(self, ngf_seg):
        pass

    def _noise(self, *shape):
        return _draw(self.noise_source, shape, self.recurrent.weight_ih.device)

    # -- recurrent text encoders (reference model.py:313-346) ------------------------------------
    def get_iteration_input(self, motion_input):
        noise = self._noise(motion_input.shape[0], self.noise_dim)          # model.py:315
        return M.dense_input(noise, motion_input, dtype=self.recurrent.in_dtype())   # cat + pad + cast in one op

    def get_gru_initial_state(self, num_samples):
        return self._noise(num_samples, self.motion_dim)                    # model.py:319

    def sample_z_motion(self, motion_input, video_len=None):
        video_len = video_len if video_len is not None else self.video_len
        num_samples = motion_input.shape[0]
        h = self.m_net(self.get_gru_initial_state(num_samples))
        # the GRU inputs of all steps (fresh noise + that step's description, model.py:313-317) do not depend on the
        # recurrence: their W_ih products are one GEMM over the time-major stack; only W_hh h runs step by step.
        # Same draw order as the reference loop (nothing else draws in between).
        if self.noise_source is None:
            noise_all = self._noise(video_len * num_samples, self.noise_dim)      # the T step noises as one time-major draw
        else:
            noise = [self._noise(num_samples, self.noise_dim) for _ in range(video_len)]
            noise_all = noise[0] if video_len == 1 else torch.cat(noise, 0)
        if motion_input.dim() == 2:
            m_all = motion_input if video_len == 1 else motion_input.repeat(video_len, 1)
        else:
            m_all = motion_input[:, :video_len].transpose(0, 1).reshape(video_len * num_samples, -1)
        gi = self.recurrent.input_gates(M.dense_input(noise_all, m_all, dtype=self.recurrent.in_dtype()))
        hs = self.recurrent.sequence(gi.view(video_len, num_samples, -1), h).transpose(0, 1)    # story-major rows (model.py:332-333), padded width
        return F.UnpadFn.apply(hs.reshape(-1, hs.shape[-1]), 0, self.motion_dim)

    def motion_content_rnn(self, motion_input, content_input):
        video_len = 1 if motion_input.dim() == 2 else self.video_len
        h = self.c_net(content_input)
        if motion_input.dim() == 2:
            motion_input = motion_input.unsqueeze(1)
        num_samples = motion_input.shape[0]
        m_all = motion_input[:, :video_len].transpose(0, 1).reshape(video_len * num_samples, -1)
        gi = self.mocornn.input_gates(m_all).view(video_len, num_samples, -1)
        hs = self.mocornn.sequence(gi, h).transpose(0, 1)
        return F.UnpadFn.apply(hs.reshape(-1, hs.shape[-1]), 0, self.content_dim)

    # -- shared trunk ---------------------------------------------------------------------------
    def _joint(self, frame_motion, zm_code, c_rows, crnn_code):
        """reference model.py:371-378 / 436-443."""
        m_image = self.image_net(frame_motion).view(-1, self.filter_num, self.image_size)
        c_filter = self.filter_net(crnn_code).view(-1, self.out_num, self.filter_num, self.filter_size)
        mc_image = self.dfn_layer([m_image, c_filter])
        return M.dense_input(zm_code, c_rows, mc_image.squeeze(1))

    def _decode(self, zmc_all):
        """reference model.py:379-405. Returns (latents, rgb NHWC, seg NHWC or None)."""
        x = F.FeatToNhwcFn.apply(self.fc(zmc_all), self.gf_dim, 4, 4)
        if not self.use_segment:
            for up in (self.upsample1, self.upsample2, self.upsample3, self.upsample4):
                x = up(x)
            return None, self.img(x), None
        if _DEC_BRANCH and zmc_all.is_cuda and self.training and os.environ.get("CPCSV_STREAMS", "1") != "0" and not torch.is_grad_enabled():
            return self._decode_two_branches(zmc_all, x)
        s = F.FeatToNhwcFn.apply(self.fc_seg(zmc_all), self.gf_dim_seg, 4, 4)
        x = F.GateFn.apply(self.seg_c(s), x)                                 # model.py:383
        s = self.upsample1_seg(s)
        x = self.upsample1(x)
        x = F.GateFn.apply(self.seg_c1(s), x)                                # model.py:387
        for ups, up in ((self.upsample2_seg, self.upsample2), (self.upsample3_seg, self.upsample3),
                        (self.upsample4_seg, self.upsample4)):
            s = ups(s)
            x = up(x)
        return None, self.img(x), self.img_seg(s)

    def _decode_two_branches(self, zmc_all, x):
        """The no-grad pass's decoder with the segmentation branch on its own stream. The two decoders only meet at the two
        gates (reference model.py:383,387: seg_c(s) and seg_c1(up1_seg(s)) scale the image branch); behind the second gate
        up2_seg..img_seg and up2..img are independent chains of GEMM -> finalize -> apply launches, each of which leaves the
        GPU partly idle at its tails. Same kernels, same order per branch, same results; the side stream's tensors come from
        that stream's pool and stay referenced until the join."""
        main = torch.cuda.current_stream()
        side = self.__dict__.get("_dec_side")
        if side is None:
            side = self.__dict__["_dec_side"] = torch.cuda.Stream()
        side.wait_stream(main)
        keep = []
        with torch.cuda.stream(side):
            s = F.FeatToNhwcFn.apply(self.fc_seg(zmc_all), self.gf_dim_seg, 4, 4)
            g0 = self.seg_c(s)
            e0 = torch.cuda.Event()
            e0.record(side)
            s = self.upsample1_seg(s)
            g1 = self.seg_c1(s)
            e1 = torch.cuda.Event()
            e1.record(side)
            keep += [g0, g1]
            for ups in (self.upsample2_seg, self.upsample3_seg, self.upsample4_seg):
                s = ups(s)
            segm = self.img_seg(s)
        main.wait_event(e0)
        x = F.GateFn.apply(g0, x)                                            # model.py:383
        x = self.upsample1(x)
        main.wait_event(e1)
        x = F.GateFn.apply(g1, x)                                            # model.py:387
        for up in (self.upsample2, self.upsample3, self.upsample4):
            x = up(x)
        rgb = self.img(x)
        main.wait_stream(side)
        del keep
        return None, rgb, segm

    # -- public API (same tuples as the reference) ------------------------------------------------
    def sample_videos(self, motion_input, content_input, seg=False):
        """reference model.py:348-423. motion (B,T,365), content (B,T,356)."""
        bs, video_len = motion_input.shape[0], motion_input.shape[1]
        content_input = content_input.reshape(-1, cfg.VIDEO_LEN * content_input.shape[2])
        r_code, r_mu, r_logvar = self.ca_net(content_input)
        c_mu = r_mu.repeat(self.video_len, 1)                                # tiled rows, quirk model.py:361
        crnn_code = self.motion_content_rnn(motion_input, r_code)            # sampled code, model.py:364
        temp = motion_input.reshape(-1, motion_input.shape[2])
        m_mu = m_logvar = temp                                               # model.py:365-366
        zm_code = self.sample_z_motion(motion_input, self.video_len)
        zmc_all = self._joint(temp, zm_code, c_mu, crnn_code)
        latents, rgb, segm = self._decode(zmc_all)
        fake = F.ToPlanarFn.apply(rgb, self.n_channels)
        fake_video = fake.view(bs, video_len, self.n_channels, self.segment_size, self.segment_size).permute(0, 2, 1, 3, 4)
        segm_video = F.ToPlanarFn.apply(segm, 1) if (segm is not None and (seg or latents is not None)) else None
        return latents, fake_video, m_mu, m_logvar, r_mu, r_logvar, (segm_video if seg else None)

    def sample_images(self, motion_input, content_input, seg=False):
        """reference model.py:426-483. motion (B,365), content (B,T,356)."""
        m_mu = m_logvar = motion_input
        content_input = content_input.reshape(-1, cfg.VIDEO_LEN * content_input.shape[2])
        c_code, c_mu, c_logvar = self.ca_net(content_input)
        crnn_code = self.motion_content_rnn(motion_input, c_mu)              # the MEAN, quirk model.py:433
        zm_code = self.sample_z_motion(motion_input, 1)
        zmc_all = self._joint(motion_input, zm_code, c_mu, crnn_code)
        latents, rgb, segm = self._decode(zmc_all)
        fake_img = F.ToPlanarFn.apply(rgb, self.n_channels)
        segm_img = F.ToPlanarFn.apply(segm, 1) if (segm is not None and seg) else None
        return latents, fake_img, m_mu, m_logvar, c_mu, c_logvar, segm_img


    def sample_both(self, st_motion, st_content, im_motion, im_content, seg=True):
        """sample_videos(st_motion, st_content) followed by sample_images(im_motion, im_content, seg) (reference
        model.py:348-483; the two calls trainer.py:295-300,367-369 always make together) with the image decoder run ONCE
        over both batches: the text / motion encoders run per call in the reference's order (same noise draws), the decoder
        (fc, fc_seg, the up-blocks, gates and output convs) sees the story frames and the images back to back, every
        BatchNorm keeping one batch per call, story first (cpcsv.runtime.row_groups). Returns the two 7-tuples."""
        bs, video_len = st_motion.shape[0], st_motion.shape[1]
        st_flat = st_content.reshape(-1, cfg.VIDEO_LEN * st_content.shape[2])
        temp = st_motion.reshape(-1, st_motion.shape[2])
        im_flat = im_content.reshape(-1, cfg.VIDEO_LEN * im_content.shape[2])
        if st_motion.is_cuda:
            self._prepack_text()
        # The text / motion encoders are four independent chains of ~30 tiny launches each (content: CA_NET -> c_net -> mocornn;
        # motion: m_net -> recurrent; per half) that meet in _joint: one after the other on one stream they are 1.3 ms of
        # launch latency with the GPU idle, at the head of the forward pass and again at the tail of the backward pass
        # (autograd runs every node on its forward stream). Host order - and with it the order of the noise draws - is
        # the sequential one; the halves carry branch roles, so the BatchNorm layers they share (m_net, c_net, image_net,
        # filter_net) update their running statistics story first, and the operand copies of the weights both halves read are
        # rebuilt before the fork.
        if self.noise_source is None and self.ca_net.noise_source is None and st_motion.is_cuda and _NOISE_BANK:
            cd, md, zd = self.content_dim, self.motion_dim, self.noise_dim
            nst_, nim_ = st_motion.shape[0], im_motion.shape[0]
            _BANK[0] = _NoiseBank(nst_ * (cd + md + self.video_len * zd) + nim_ * (cd + md + zd), st_motion.device)
        try:
            return self._sample_both(st_motion, st_content, im_motion, im_content, seg, bs, video_len, st_flat, temp, im_flat)
        finally:
            _BANK[0] = None

    def _sample_both(self, st_motion, st_content, im_motion, im_content, seg, bs, video_len, st_flat, temp, im_flat):
        if TP.supported(self, st_motion, st_content, im_motion, im_content):
            # the text / motion encoders of both calls as ~10 stage launches (cpcsv/textpath.py) instead of ~80 per-layer ones
            dev = st_motion.device
            zmc, r_mu, r_logvar, c_mu, c_logvar = TP.text_path(
                self, st_motion, st_flat, im_motion, im_flat,
                lambda shape: _draw(self.ca_net.noise_source, tuple(shape), dev), lambda shape: _draw(self.noise_source, tuple(shape), dev))
            return self._decode_both(zmc, bs * video_len, im_motion.shape[0], bs, video_len, seg, temp, im_motion, r_mu, r_logvar, c_mu, c_logvar)
        par = self._text_streams(st_motion)
        if par is None:
            st_c = st_z = im_c = im_z = contextlib.nullcontext()
            join = lambda: None
        else:
            main = torch.cuda.current_stream()
            for s_ in par:
                s_.wait_stream(main)
            on = lambda s_, i, role: _Both(torch.cuda.stream(s_), branch(i, role))
            st_c, st_z, im_c, im_z = on(par[0], 1, "first"), on(par[1], 1, "first"), on(par[2], 2, "second"), on(par[3], 2, "second")

            def join():
                par[0].wait_stream(par[1])
                par[2].wait_stream(par[3])
        with st_c:
            r_code, r_mu, r_logvar = self.ca_net(st_flat)
            crnn_st = self.motion_content_rnn(st_motion, r_code)                 # sampled code, model.py:364
        with st_z:
            zm_st = self.sample_z_motion(st_motion, self.video_len)
        with im_c:
            _, c_mu, c_logvar = self.ca_net(im_flat)
            crnn_im = self.motion_content_rnn(im_motion, c_mu)                   # the MEAN, quirk model.py:433
        with im_z:
            zm_im = self.sample_z_motion(im_motion, 1)
        join()
        with st_c:
            zmc_st = self._joint(temp, zm_st, r_mu.repeat(self.video_len, 1), crnn_st)      # tiled rows, quirk model.py:361
        with im_c:
            zmc_im = self._joint(im_motion, zm_im, c_mu, crnn_im)
        if par is not None:
            main.wait_stream(par[0])
            main.wait_stream(par[2])
            # (side-stream tensors read on the main stream: their blocks return to the side streams' pools, whose next use is
            # ordered behind the next pass's fork from the main stream - no early reuse)
        nst, nim = zmc_st.shape[0], zmc_im.shape[0]
        return self._decode_both(torch.cat((zmc_st, zmc_im), 0), nst, nim, bs, video_len, seg, temp, im_motion, r_mu, r_logvar, c_mu, c_logvar)

    def _decode_both(self, zmc_all, nst, nim, bs, video_len, seg, temp, im_motion, r_mu, r_logvar, c_mu, c_logvar):
        """ONE decoder pass over the story frames | images (row groups: every BatchNorm keeps one batch per call, story first)."""
        with row_groups((nst, nim)):
            latents, rgb, segm = self._decode(zmc_all)
        st_fake, im_fake = F.ToPlanarSplitFn.apply(rgb, self.n_channels, (nst, nim))
        st_video = st_fake.view(bs, video_len, self.n_channels, self.segment_size, self.segment_size).permute(0, 2, 1, 3, 4)
        se_img = F.ToPlanarFn.apply(segm[nst:], 1) if (segm is not None and seg) else None
        lat_st = lat_im = None
        if latents is not None:
            lat_st = tuple(tuple(t[:nst] for t in grp) for grp in latents)
            lat_im = tuple(tuple(t[nst:] for t in grp) for grp in latents)
        return ((lat_st, st_video, temp, temp, r_mu, r_logvar, None),
                (lat_im, im_fake, im_motion, im_motion, c_mu, c_logvar, se_img))


    def _text_streams(self, like):
        """Four side streams for the encoder chains of sample_both, or None: CPCSV_TEXT_STREAMS=0, CPU tensors, an injected noise
        source (parity runs stay on one stream), or a differentiable pass (forking that one too was measured at +0.5 ms/step in round 3 -
        its forward hides behind the critic updates anyway, and the backward pays a cross-stream dependency per node - and the switch
        for it is gone)."""
        if not _TEXT_STREAMS or not like.is_cuda or self.noise_source is not None or self.ca_net.noise_source is not None:
            return None
        if torch.is_grad_enabled():
            return None
        uses = self._text_uses_list()
        grad = torch.is_grad_enabled()
        for lay in uses:
            w = lay.holder.master()
            lay.packs(w, M.L.F32 if lay.compute_f32 else dcode(), "both" if (grad and w.requires_grad) else "fwd")
        st = self.__dict__.get("_text_side")
        if st is None:
            st = self.__dict__["_text_side"] = [torch.cuda.Stream() for _ in range(4)]
        return st

    def _text_uses_list(self):
        uses = self.__dict__.get("_text_uses")
        if uses is None:
            self.__dict__["_text_uses"] = uses = [lay for lay in self._text_layers()]
        return uses

    def _prepack_text(self):
        """All stale operand copies of the text / motion encoders' nine small fp32 layers in ONE launch at the head of a pass (they
        are rewritten by the multi-tensor Adam launch, i.e. stale once per step): forward layouts always, the data-gradient
        layouts too while training (the differentiable pass of the same step reads them). CPCSV_PREPACK_TEXT=0: per layer, lazily."""
        if not _PREPACK_TEXT:
            return
        M.prepack_dense(self._text_uses_list(), ("fwd", "bwd") if self.training else ("fwd",))

    def _text_layers(self):
        yield M._layer_for(self.ca_net.fc, None, M.L.ACT_RELU, 0, out_mode="f32pad")
        for seq in (self.m_net, self.c_net, self.image_net, self.filter_net):
            for lay in seq._plan():
                if isinstance(lay, M.KernelLayer):
                    yield lay
        for cell in (self.recurrent, self.mocornn):
            for lay in cell._layers():
                yield lay


class _Both:
    """two context managers as one"""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def __enter__(self):
        self.a.__enter__()
        self.b.__enter__()

    def __exit__(self, *e):
        self.b.__exit__(*e)
        self.a.__exit__(*e)


STAGE1_G = StoryGAN


class R2Plus1dStem(nn.Sequential):
    """Stem of the order critic (reference model.py:15-29): SN-Conv3d(3,45,(1,7,7),s(1,2,2),p(0,3,3)) + BN3d + ReLU,
    SN-Conv3d(45,64,1x1x1, padding (1,0,0): T grows by 2) + BN3d + ReLU. Parameter holders only; VideoEncoder runs them."""

    def __init__(self):
        super().__init__(M.Conv3d(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)), M.BatchNorm3d(45), nn.ReLU(inplace=True),
                         M.Conv3d(45, 64, (1, 1, 1), (1, 1, 1), (1, 0, 0)), M.BatchNorm3d(64), nn.ReLU(inplace=True))


class VideoEncoder(nn.Module):
    """Order-consistency critic on whole stories (reference model.py:99-210; enabled by cfg.USE_SEQ_CONSISTENCY):
    (B,3,T,64,64) -> (B,1) logit. Spatial (1,3,3)/s(1,2,2) convs run as ordinary 3x3 stride-2 convs over the B*T frames,
    temporal (3,1,1)/s(2,1,1) convs as (3,1)-kernel (2,1)-stride convs over [B][T][H*W] "images" - both through the
    gather-GEMM with BatchNorm statistics from its epilogue (BatchNorm3d = statistics over all rows); the 7x7 stem as a
    patch matrix (cpcsv_im2col) times the master viewed [45][147]; the 1x1x1 conv as a dense layer over the channel axis
    of the time-padded frames. Module tree and state_dict keys equal the reference's."""

    def __init__(self):
        super().__init__()
        sp = lambda ci, co: M.Conv3d(ci, co, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        tm = lambda ci, co: M.Conv3d(ci, co, (3, 1, 1), (2, 1, 1), (1, 0, 0))
        block = [R2Plus1dStem()]
        for mk, ci, co in ((sp, 64, 128), (tm, 128, 128), (sp, 128, 128), (tm, 128, 256), (sp, 256, 256), (tm, 256, 512),
                           (sp, 512, 512), (tm, 512, 512)):
            block += [mk(ci, co), M.BatchNorm3d(co), nn.LeakyReLU(0.2)]
        self.pool = nn.Identity()                      # nn.AdaptiveAvgPool3d(1) in the reference: no parameters
        self.story_encoder = nn.Sequential(*block)
        self.detector = M.FusedSequential(M.Linear(512, 128, spectral=True), M.BatchNorm1d(128), nn.ReLU(),
                                          M.Linear(128, 1, spectral=True), out_mode="f32")
        self._lay = None

    def _layers(self):
        if self._lay is None:
            L = M.L
            stem = self.story_encoder[0]
            mk_dense = lambda conv, bn, act, cin, name: M.KernelLayer(conv, bn, act, 0, "dense", cin, conv.cout, 1, 1, None, None, "T", name)
            lay = {"stem": mk_dense(stem[0], stem[1], L.ACT_RELU, 3 * 49, "OrderCritic.stem7x7"),
                   "point": mk_dense(stem[3], stem[4], L.ACT_RELU, 45, "OrderCritic.point")}
            for l_ in lay.values():
                l_.compute_f32 = False                 # 61k-86k rows: these are real GEMMs, keep the compute dtype
            kids = list(self.story_encoder.children())[1:]
            lay["tower"] = [M._layer_for(kids[i], kids[i + 1], L.ACT_LRELU, 0) for i in range(0, len(kids), 3)]
            self._lay = lay
        return self._lay

    def forward(self, story):
        lay = self._layers()
        b, c, t, h, w = story.shape
        x = F.ToNhwcFn.apply(story, tdtype())                          # [B*T, H, W, 8]
        x = lay["stem"](F.Im2colFn.apply(x, 3, 7, 2, 3))               # [B*T*(H/2)*(W/2), 48]
        hw = (h // 2) * (w // 2)
        x5 = x.view(b, t, hw * x.shape[1])
        zero = torch.zeros_like(x5[:, :1])
        x = torch.cat((zero, x5, zero), 1).view(b * (t + 2) * hw, -1)  # time padding (1,0,0) of the 1x1x1 conv
        x = lay["point"](x)                                            # [B*(T+2)*hw, 64]
        t, h, w = t + 2, h // 2, w // 2
        x = x.view(b * t, h, w, x.shape[1])
        for conv_lay in lay["tower"]:
            if conv_lay.holder.temporal:
                y = conv_lay(x.view(b, t, h * w, x.shape[-1]))         # [B, T', H*W, C]
                t = y.shape[1]
                x = y.view(b * t, h, w, y.shape[-1])
            else:
                x = conv_lay(x)
                h, w = x.shape[1], x.shape[2]
        p = t * h * w
        x = F.MeanTFn.apply(x.view(b * p, 1, 1, x.shape[-1]), p)      # AdaptiveAvgPool3d(1)
        return self.detector(x.view(b, -1))                            # (B, 1) fp32


def _tower(cin, ndf, first_spectral):
    """Four conv4x4 s2 p1 stages (reference model.py:498-514, 540-556, 582-598)."""
    return M.FusedSequential(
        M.Conv2d(cin, ndf, 4, 2, 1, bias=False, spectral=first_spectral),
        nn.LeakyReLU(0.2, inplace=True),
        M.Conv2d(ndf, ndf * 2, 4, 2, 1, bias=False, spectral=True),
        M.BatchNorm2d(ndf * 2),
        nn.LeakyReLU(0.2, inplace=True),
        M.Conv2d(ndf * 2, ndf * 4, 4, 2, 1, bias=False, spectral=True),
        M.BatchNorm2d(ndf * 4),
        nn.LeakyReLU(0.2, inplace=True),
        M.Conv2d(ndf * 4, ndf * 8, 4, 2, 1, bias=False, spectral=True),
        M.BatchNorm2d(ndf * 8),
        nn.LeakyReLU(0.2, inplace=True))


class _Critic(nn.Module):
    in_channels = 3
    first_spectral = False

    def __init__(self, use_categories=True):
        super().__init__()
        self.df_dim = cfg.GAN.DF_DIM
        self.ef_dim = cfg.GAN.CONDITION_DIM
        self.text_dim = cfg.TEXT.DIMENSION
        self.label_num = cfg.LABEL_NUM
        self.define_module(use_categories)

    def define_module(self, use_categories):
        ndf, nef = self.df_dim, self.ef_dim
        self.encode_img = _tower(self.in_channels, ndf, self.first_spectral)
        self.seq_consisten_model = None
        self.get_cond_logits = D_GET_LOGITS(ndf, nef + self.text_dim + self.label_num)
        self.get_uncond_logits = None
        self.cate_classify = M.HeadConv2d(ndf * 8, self.label_num, 4, 4, 1, bias=False) if use_categories else None

    def forward(self, image):
        feat = self.encode_img(F.ToNhwcFn.apply(image, tdtype()))
        return feat.permute(0, 3, 1, 2)          # (N, 8*ndf, 4, 4) view, like the reference's return

    def encode_pair(self, real, fake):
        """netD(real) and netD(fake) of a critic update (reference miscc/utils.py:70-71) as ONE pass over both batches:
        same launches, twice the rows; BatchNorm statistics and the spectral-norm iteration stay per batch, real first.
        Returns the (2N, 8*ndf, 4, 4) features, real rows first."""
        x = F.ToNhwcCatFn.apply(tdtype(), real, fake)
        nr = x.shape[0] * real.shape[0] // (real.shape[0] + fake.shape[0])
        with row_groups((nr, x.shape[0] - nr)):
            feat = self.encode_img(x)
        return self._pool(feat, real).permute(0, 3, 1, 2)

    def _pool(self, feat, like):
        return feat


class STAGE1_D_IMG(_Critic):
    """reference model.py:487-527"""


class STAGE1_D_SEG(_Critic):
    """reference model.py:529-569"""
    in_channels = 1


class STAGE1_D_STY_V2(_Critic):
    """reference model.py:571-618: frames folded into the batch, features averaged over T."""
    first_spectral = True

    def __init__(self):
        super().__init__(use_categories=False)
        if cfg.USE_SEQ_CONSISTENCY:                            # reference model.py:599-601
            self.seq_consisten_model = VideoEncoder()

    def forward(self, story):
        n, c, video_len, w, h = story.shape
        frames = F.ToNhwcFn.apply(story, tdtype())            # (N*T, H, W, Cs): the permute/contiguous of :612-613
        feat = self.encode_img(frames)
        feat = F.MeanTFn.apply(feat, video_len)               # :616-617
        return feat.permute(0, 3, 1, 2)

    def _pool(self, feat, like):
        return F.MeanTFn.apply(feat, like.shape[2])           # frames -> stories, both batches at once
