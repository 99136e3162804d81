"""trainer.py — GANTrainer on the MI355X-native path.

Drop-in for the reference's trainer.GANTrainer (reference trainer.py:42-485): same constructor
(`GANTrainer(output_dir, args, ratio)`), `load_network_stageI()`, `train(imageloader, storyloader,
testloader, stage)`, same batch-dict schema, same update order, LR schedule and checkpoint files.
The loop body (reference :252-416) is `train_step`, callable on its own with device-resident
batches (bench.py, tests).

What differs, and why (results unchanged):
  * one process per GPU; gradients of each optimiser are mean-all-reduced over RCCL right before its
    step (cpcsv.dist) instead of the reference's single-process data_parallel on the critics;
  * the four Adam optimisers are single-launch multi-tensor kernels (cpcsv.optim.FusedAdam);
  * the critic weight gradients that the reference computes during the G step and then throws away
    (zero_grad at :313-317 clears them before any use) are not computed;
  * no device->host sync inside the step: accuracies/losses stay device scalars until logged;
  * eval hooks (FID/FVD/SSIM, reference :160-185) need TensorFlow / pytorch_ssim and are outside the hot path
    (SURVEY §2 rows 13-15): requesting them raises. The epoch-end sample (reference :437-444) IS run: it is a
    train-mode pass that draws noise and advances the BatchNorm running statistics.
"""
from __future__ import print_function

import os
import time
from shutil import copyfile

import numpy as np
import torch

from cpcsv import dist as cdist
from cpcsv import graphs
from cpcsv import ingest
from cpcsv import runtime
from cpcsv import spectral
from cpcsv._lib import F32 as L_F32
from cpcsv.optim import FusedAdam
from miscc.config import cfg
from miscc.utils import (KL_loss, compute_discriminator_loss, compute_generator_loss, count_param, mkdir_p,
                         mse_loss, save_image_results, save_model, save_story_results, weights_init)

try:  # the reference logs with tensorboardX (trainer.py:34,80); optional here
    from tensorboardX import SummaryWriter
except Exception:  # pragma: no cover
    SummaryWriter = None


_FUSED_MIN_NUMEL = int(os.environ.get("CPCSV_FUSED_MIN_NUMEL", str(1 << 16)))    # smaller weights stay on the multi-tensor Adam path


class _ScalarLog(object):
    """Minimal stand-in for SummaryWriter: keeps scalars as device tensors, reads them back in one
    batch on flush() so logging never forces a per-step sync (reference :357-360 syncs every step)."""

    def __init__(self, log_dir=None):
        self.rows, self.log_dir, self.images = [], log_dir, []

    def add_scalar(self, key, value, step):
        self.rows.append((key, value, step))

    def add_image(self, tag, img, step=None, **k):
        self.images.append((tag, tuple(getattr(img, "shape", ())), step))      # (the sheet itself is not kept)

    def flush(self):
        out = [(k, float(v), s) for k, v, s in self.rows]
        self.rows = []
        return out


class GANTrainer(object):
    def __init__(self, output_dir, args, ratio=1.0):
        if cfg.TRAIN.FLAG and output_dir is not None:
            output_dir = "{}/".format(output_dir)
            self.model_dir = os.path.join(output_dir, 'Model')
            self.image_dir = os.path.join(output_dir, 'Image')
            self.log_dir = os.path.join(output_dir, 'log')
            self.test_dir = os.path.join(output_dir, 'Test')
            for d in (self.model_dir, self.image_dir, self.log_dir, self.test_dir):
                mkdir_p(d)
            here = os.path.dirname(os.path.abspath(__file__))
            if not os.path.exists(os.path.join(self.model_dir, 'model.py')):     # reference :55-61
                cfg_file = getattr(args, 'cfg_file', None)
                if cfg_file and os.path.exists(cfg_file):
                    copyfile(cfg_file, output_dir + 'setting.yml')
                copyfile(os.path.join(here, 'cascade_model.py' if cfg.CASCADE_MODEL else 'model.py'), output_dir + 'model.py')
                copyfile(os.path.join(here, 'trainer.py'), output_dir + 'trainer.py')
        else:
            self.model_dir = self.image_dir = self.log_dir = self.test_dir = None
        self.video_len = cfg.VIDEO_LEN
        self.max_epoch = cfg.TRAIN.MAX_EPOCH
        self.snapshot_interval = cfg.TRAIN.SNAPSHOT_INTERVAL
        self.gpus = [int(ix) for ix in str(cfg.GPU_ID).split(',')]              # reference :67-71
        self.num_gpus = len(self.gpus)
        self.imbatch_size = cfg.TRAIN.IM_BATCH_SIZE * self.num_gpus
        self.stbatch_size = cfg.TRAIN.ST_BATCH_SIZE * self.num_gpus
        self.ratio = ratio
        self.con_ckpt = getattr(args, 'continue_ckpt', None)
        self.rank, self.world, self.local_rank = cdist.init_from_env()
        self.exchange = self.world > 1 or cdist.force_exchange()     # gradients are exchanged before every optimiser step
        if not torch.cuda.is_available():
            raise RuntimeError("GANTrainer runs on MI355X GPUs only; no CPU fallback exists in the product path")
        torch.cuda.set_device(self.local_rank if self.world > 1 else self.gpus[0])
        self.device = torch.device('cuda', torch.cuda.current_device())
        if self._streams_on() and os.environ.get("CPCSV_BIND_STREAMS", "1") != "0":
            # Bind the main stream and the three critic streams to hardware queues NOW: the HIP runtime hands its 4 queues out round
            # robin at a stream's FIRST SUBMISSION, and the process group's first collective (the replica broadcast in setup())
            # creates RCCL's and ProcessGroupNCCL's streams. Bound after those, the main stream and a critic's stream landed on one
            # queue: the generator's forward could not start before that critic's update had drained (data-parallel runs: +1.0 ms
            # per step, tools/phase_times.py under CPCSV_FORCE_EXCHANGE=1: forward starts +0.16 ms behind the no-grad pass with
            # this, +3.5 ... +4.5 without; profiles/r05_rccl_rehearsal.txt). Harmless without a process group.
            t = torch.zeros(8, device=self.device)
            t.add_(1.0)
            for key in ("se", "im", "st"):
                with torch.cuda.stream(self._side_stream(key)):
                    t.add_(1.0)
            torch.cuda.synchronize()
            if self.exchange:
                # ... and the ONE communication stream (cpcsv/dist.py): fifth in the round robin, it shares a hardware queue with one
                # of the four above; CPCSV_COMM_QUEUE = number of throw-away streams submitted first (0: the main stream's queue,
                # 1 / 2 / 3: the se / im / st critic's). World-1 rehearsal at bench widths (profiles/r06_rccl_rehearsal.txt, fp32 wire):
                # 15.25 / 15.46 / 14.50 / 16.80 ms per step for 0 / 1 / 2 / 3 against 13.34 without the exchange
                self._comm_spacers = cdist.bind_comm_stream(int(os.environ.get("CPCSV_COMM_QUEUE", "2")))
        self._logger = (SummaryWriter(self.log_dir) if (SummaryWriter and self.log_dir and self.rank == 0)
                        else _ScalarLog(self.log_dir))
        self.nets = None

    # ---------------------------------------------------------------- networks (reference :82-140)
    def load_network_stageI(self):
        if cfg.CASCADE_MODEL:
            from cascade_model import StoryGAN, STAGE1_D_IMG, STAGE1_D_STY_V2, STAGE1_D_SEG
        else:
            from model import StoryGAN, STAGE1_D_IMG, STAGE1_D_STY_V2, STAGE1_D_SEG
        netG = StoryGAN(self.video_len)
        netG.apply(weights_init)
        netD_im = STAGE1_D_IMG()
        netD_im.apply(weights_init)
        netD_st = STAGE1_D_STY_V2()
        netD_st.apply(weights_init)
        netD_se = None
        if cfg.SEGMENT_LEARNING:
            netD_se = STAGE1_D_SEG()
            netD_se.apply(weights_init)
        if self.rank == 0:
            total = count_param(netG) + count_param(netD_im) + count_param(netD_st) + (count_param(netD_se) if netD_se else 0)
            print('The total parameter is : {}M, netG:{}M, netD_im:{}M, netD_st:{}M'.format(
                total // 1e6, count_param(netG) // 1e6, count_param(netD_im) // 1e6, count_param(netD_st) // 1e6))
        if cfg.NET_G != '':
            netG.load_state_dict(torch.load(cfg.NET_G, map_location='cpu'))
            print('Load from: ', cfg.NET_G)
        if self.con_ckpt:                                                        # reference :121-131
            print('Continue training from epoch {}'.format(self.con_ckpt))
            netG.load_state_dict(torch.load('{}/netG_epoch_{}.pth'.format(self.model_dir, self.con_ckpt), map_location='cpu'))
            netD_im.load_state_dict(torch.load('{}/netD_im_epoch_last.pth'.format(self.model_dir), map_location='cpu'))
            netD_st.load_state_dict(torch.load('{}/netD_st_epoch_last.pth'.format(self.model_dir), map_location='cpu'))
            if netD_se is not None:
                netD_se.load_state_dict(torch.load('{}/netD_se_epoch_last.pth'.format(self.model_dir), map_location='cpu'))
        for n in (netG, netD_im, netD_st, netD_se):
            if n is not None:
                n.to(self.device)
                cdist.broadcast_module(n)          # identical replicas; SN u/v then stay identical without comm
        return netG, netD_im, netD_st, netD_se

    def sample_real_image_batch(self):
        if self.imagedataset is None:
            self.imagedataset = enumerate(self.imageloader)
        batch_idx, batch = next(self.imagedataset)
        b = ingest.to_device_batch(batch, self.device, feeder=True)       # pinned staging + copy stream; uint8 frames are normalised on the device
        if batch_idx == len(self.imageloader) - 1:
            self.imagedataset = enumerate(self.imageloader)
        return b

    # ---------------------------------------------------------------- set-up (reference :192-228)
    def setup(self, nets=None):
        """Create nets (unless given), labels and the four optimisers. Called by train(); public for bench/tests."""
        self.nets = nets if nets is not None else self.load_network_stageI()
        netG, netD_im, netD_st, netD_se = self.nets
        dev = self.device
        self.im_real_labels = torch.ones(self.imbatch_size, device=dev)
        self.im_fake_labels = torch.zeros(self.imbatch_size, device=dev)
        self.st_real_labels = torch.ones(self.stbatch_size, device=dev)
        self.st_fake_labels = torch.zeros(self.stbatch_size, device=dev)
        self.generator_lr = cfg.TRAIN.GENERATOR_LR
        self.discriminator_lr = cfg.TRAIN.DISCRIMINATOR_LR
        adam = lambda net, lr: FusedAdam([p for p in net.parameters() if p.requires_grad], lr=lr, betas=(0.5, 0.999))
        self.im_optimizerD = adam(netD_im, cfg.TRAIN.DISCRIMINATOR_LR)
        self.st_optimizerD = adam(netD_st, cfg.TRAIN.DISCRIMINATOR_LR)
        self.se_optimizerD = adam(netD_se, cfg.TRAIN.DISCRIMINATOR_LR) if netD_se is not None else None
        self.optimizerG = adam(netG, cfg.TRAIN.GENERATOR_LR)
        # gradient payload on the wire: the compute dtype - bf16 training exchanges bf16 (half the bytes over xGMI, pipelined chunk by
        # chunk with the layer updates like the fp32 wire: cpcsv.dist.GradBucket.reduce_extra_async), fp32 mode exchanges fp32 like
        # the reference's reduction. CPCSV_GRAD_COMM=fp32|bf16 overrides.
        payload = os.environ.get("CPCSV_GRAD_COMM") or ("fp32" if runtime.dcode() == L_F32 else "bf16")
        self._buckets = {k: cdist.GradBucket(n.parameters(), payload=payload).adopt() for k, n in
                         (("G", netG), ("im", netD_im), ("st", netD_st), ("se", netD_se)) if n is not None}
        self._opt_of = {"G": self.optimizerG, "im": self.im_optimizerD, "st": self.st_optimizerD, "se": self.se_optimizerD}
        for opt in self._opt_of.values():
            if opt is not None and self.exchange:
                opt.inline = False         # the gradient all-reduce has to come between the backward pass and any update
        # all spectral-norm power iterations of a phase in one launch triple per round (cpcsv/spectral.py)
        self._sn_plans = {}
        if os.environ.get("CPCSV_SN_PLAN", "1") != "0":
            self._sn_plans = {k: p for k, p in (("im", spectral.plan_for_critic(netD_im)), ("st", spectral.plan_for_critic(netD_st)),
                                                ("se", spectral.plan_for_critic(netD_se))) if p is not None}
        if os.environ.get("CPCSV_FUSED_UPDATE", "1") != "0":
            for key, net, opt in (("G", netG, self.optimizerG), ("im", netD_im, self.im_optimizerD),
                                  ("st", netD_st, self.st_optimizerD), ("se", netD_se, self.se_optimizerD)):
                if net is not None:
                    self._attach_deferred_updates(net, opt, self._buckets[key])
        return self.nets

    def _attach_deferred_updates(self, net, opt, bucket):
        """Big conv / dense weights leave the per-call unpack -> multi-tensor Adam -> re-pack pipeline: their backward only
        accumulates, and the optimiser step runs one fused launch per layer (cpcsv.optim.FusedAdam.attach_layer,
        include/cpcsv_hip.h cpcsv_layer_update). All accumulators of a net live in ONE flat buffer that the gradient
        bucket zeroes and all-reduces together with the ordinary gradients. Spectral-normed layers take part when they
        feed a BatchNorm (closed-form <G,W>) and the run is single-GPU (their per-call coefficients are rank-local)."""
        from cpcsv import modules as M
        from cpcsv import _lib as L
        dt = runtime.dcode()
        layers, seen = [], set()
        for m in net.modules():
            if isinstance(m, M.FusedSequential):
                for lay in m._plan():
                    if isinstance(lay, M.KernelLayer):
                        layers.append(lay)
                        seen.add(id(lay.holder))
        for m in net.modules():
            if type(m) is M.Conv2d and id(m) not in seen:              # stand-alone convs (seg_c, seg_c1)
                layers.append(M._layer_for(m, None, L.ACT_NONE, 0))
        picked = []
        for lay in layers:
            h = lay.holder
            w = h.master() if hasattr(h, "master") else None
            if w is None or not w.requires_grad or lay.compute_f32 or lay.tapmap is not None or w.numel() < _FUSED_MIN_NUMEL:
                continue
            if lay.kind == "conv" and lay.cout <= 4:
                continue                                                   # streaming thin layers keep the simple path
            if getattr(h, "spectral", False) and lay.bn is None:
                continue
            if any(l2 is not lay and l2.holder is h for l2 in layers):
                continue                                                   # one master, several operand layouts
            picked.append((lay, w))
        if not picked:
            return
        sizes = [lay.cout * lay.slices * lay.cin_s for lay, _ in picked]
        acc = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        off = 0
        for (lay, w), n in zip(picked, sizes):
            lay._g = acc[off:off + n].view(lay.cout, lay.slices * lay.cin_s)
            lay._g_span = (len(bucket.extra), off, off + n)     # (index of `acc` in bucket.extra, element range): the exchange's chunk map
            off += n
            lay.fused_dt = dt
            with torch.no_grad():
                lay.packs(w, dt, "both")          # allocates the operand buffers the fused launch rewrites (and zeroes their pads)
            opt.attach_layer(lay, w)
        bucket.extra.append(acc)
        bucket.__dict__.setdefault("fused_layers", []).extend(lay for lay, _ in picked)      # (GradBucket.zero: store-first accumulators)
        # data-parallel runs: a spectral-normed layer's per-call <G, W> (closed form out of cpcsv_bn_bwd_apply; it scales the rank-1
        # term of its deferred update) must be the MEAN over the ranks like the gradient itself - the values live in 4 floats per
        # layer behind the net's flat gradient buffer and travel with its all-reduce (cpcsv.dist.GradBucket.adopt(scalars=))
        sn = [lay for lay, _ in picked if getattr(lay.holder, "spectral", False)] if self.exchange else []
        bucket.adopt(retired=[w for _, w in picked], scalars=4 * len(sn))      # their .grad views leave the zeroed / all-reduced part of the buffer
        for i, lay in enumerate(sn):
            lay.gw_slots = bucket.scalars[4 * i:4 * i + 4]

    def _exchange_and_step(self, key, opt):
        """Gradient mean over the ranks + optimiser step of one net, on the current stream. Single rank: just the step. Several
        ranks: the small flat buffer is mean-reduced in one collective; the layer accumulators (99 % of the bytes) go out in
        chunks whose fused layer updates start as soon as each chunk has landed (cpcsv.dist.GradBucket.reduce_extra_async),
        with the 1/world folded into the update kernel."""
        if not self.exchange:
            opt.step()
            return
        opt.flush_stashes()
        bucket = self._buckets[key]
        pending = bucket.reduce_extra_async() if os.environ.get("CPCSV_COMM_PIPELINE", "1") != "0" else None
        if pending is None:
            bucket.allreduce_mean()
            opt.step()
        else:
            bucket.allreduce_mean(skip_extra=True)
            opt.step(pending=pending, gscale=1.0 / self.world, wire_of=bucket.wire_of)

    def _side_stream(self, key):
        """One HIP stream per critic (CPCSV_STREAMS=0 runs everything on the current stream)."""
        if os.environ.get("CPCSV_STREAMS", "1") == "0":
            return torch.cuda.current_stream()
        st = getattr(self, "_streams", None)
        if st is None:
            st = self._streams = {}
        if key not in st:
            st[key] = torch.cuda.Stream()      # (stream priorities were measured in round 4: +1.0 ... +5.7 ms per step)
        return st[key]

    def _nograd_fakes(self, st_m, st_c, im_m, im_c):
        """The generator pass that makes the critics' fakes (reference :295-300): no autograd, no dependence on the
        critics, fixed shapes - ~450 launches. Captured once as a HIP graph and replayed (cpcsv/graphs.py; fresh
        noise every replay through torch's graph-safe RNG). CPCSV_NOGRAD_GRAPH=0, an injected noise source (parity
        tests) or a changed batch shape run the eager pass."""
        netG = self.nets[0]
        gc_ = self.__dict__.get("_ng")
        if gc_ is None:
            from cpcsv import modules as M

            def eager(a, b, c, d):
                import miscc.utils as MU
                if MU.BATCH_PASSES and hasattr(netG, "sample_both"):
                    # story half and image half decoded together (model.StoryGAN.sample_both): one set of launches
                    with torch.no_grad():
                        (_, st_fake, _, _, c_mu, _, _), (_, im_fake, _, _, cim_mu, _, se_fake) = netG.sample_both(a, b, c, d, seg=True)
                    return st_fake, c_mu, im_fake, cim_mu, se_fake
                two = self._streams_on() and graphs.env_on("CPCSV_G_BRANCHES") and self.__dict__.get("_g_packs")
                with torch.no_grad():
                    if not two:
                        log, M.PACK_LOG = M.PACK_LOG, ([] if M.PACK_LOG is None else M.PACK_LOG)
                        try:
                            _, st_fake, _, _, c_mu, _, _ = netG.sample_videos(a, b)
                            _, im_fake, _, _, cim_mu, _, se_fake = netG.sample_images(c, d, seg=True)
                        finally:
                            if log is None:                 # remember which layers repack after an optimiser step
                                if M.PACK_LOG and not self.__dict__.get("_g_packs"):
                                    self._g_packs = list(M.PACK_LOG)
                                M.PACK_LOG = None
                        return st_fake, c_mu, im_fake, cim_mu, se_fake
                    # two halves on two streams: all weight repacks first (both halves read them), then fork
                    cur, s2, s3 = torch.cuda.current_stream(), self._side_stream("g2"), self._side_stream("g3")
                    s3.wait_stream(cur)
                    for layer, weight, dt, _ in self._g_packs:
                        layer.packs(weight, dt, "fwd")
                    with torch.cuda.stream(s3):             # the data-gradient layouts are not needed before the backward
                        for layer, weight, dt, _ in self._g_packs:
                            layer.packs(weight, dt, "bwd")
                    s2.wait_stream(cur)
                    with runtime.branch(1, "first"):
                        _, st_fake, _, _, c_mu, _, _ = netG.sample_videos(a, b)
                    with torch.cuda.stream(s2), runtime.branch(2, "second"):
                        _, im_fake, _, _, cim_mu, _, se_fake = netG.sample_images(c, d, seg=True)
                    cur.wait_stream(s2)
                    cur.wait_stream(s3)
                return st_fake, c_mu, im_fake, cim_mu, se_fake
            gc_ = self._ng = graphs.GraphedCall(eager, "the no-grad generator pass", bn_owner=netG,
                                                enabled=lambda: graphs.env_on("CPCSV_NOGRAD_GRAPH") and netG.noise_source is None)
        return gc_(st_m, st_c, im_m, im_c)

    def _critic_real(self, key, net, real_imgs):
        """netD(real_imgs) (reference miscc/utils.py:70) ahead of time: it needs nothing from the generator, so it runs
        on the critic's stream while the main stream is still producing the fakes. Captured as its own graph; the
        graph of `_critic_backward` (same memory pool) back-propagates into it."""
        calls = self.__dict__.setdefault("_cr", {})
        gc_ = calls.get(key)
        if gc_ is None:
            gc_ = calls[key] = graphs.GraphedCall(lambda real: net(real), "the %s critic's real-image pass" % key, bn_owner=net,
                                                  stream=self._side_stream(key), enabled=lambda: self._critic_graph_on(key))
        return gc_(real_imgs)

    def _critic_graph_on(self, key):
        rest = self.__dict__.get("_cg", {}).get(key)
        # (with the order critic on, create_random_shuffle's host decisions reach the captured pass through persistent device
        # index tensors refreshed before every replay: miscc.utils.ShufflePlanBuffers)
        return graphs.env_on("CPCSV_CRITIC_GRAPH") and self._streams_on() and not (rest is not None and rest.off)

    def _critic_backward(self, key, net, a, tag, real_features):
        """One critic's zero_grad + losses + backward (reference :313-346 without the optimiser step): fixed shapes,
        ~300 launches, captured once per critic on that critic's stream and replayed. The optimiser step (and the
        gradient all-reduce in front of it) stays outside the graph. CPCSV_CRITIC_GRAPH=0 keeps it eager."""
        gpus = self.gpus
        calls = self.__dict__.setdefault("_cg", {})
        feats = self.__dict__.setdefault("_feat", {})
        holds = self.__dict__.setdefault("_dhold", {})
        feats[key] = real_features            # eager tensor, or the static output of the real-image graph
        gc_ = calls.get(key)
        if gc_ is None:
            def eager(real, fake, real_labels, fake_labels, cate, cond):
                self._buckets[key].zero()                      # net.zero_grad(), reference :313-317
                self._opt_of[key].prepare_step()
                errD, e_r, e_w, e_f, accD, cons = compute_discriminator_loss(net, real, fake, real_labels, fake_labels, cate, cond,
                                                                             gpus, real_features=feats[key])
                errD.backward(self._root_grad(errD))
                res = {tag + '/loss': errD.detach(), tag + '/real': e_r, tag + '/wrong': e_w, tag + '/fake': e_f}
                if net.seq_consisten_model is not None:
                    res[tag + '/order'] = cons                # reference trainer.py:360
                if key != "st":
                    res['Accuracy/%s_D' % key] = accD
                # the logged scalars leave the graph's memory pool (one stack + one copy, captured with the rest): the pool is
                # shared with the real-image graph, whose look-ahead replay for the NEXT step (train_step(next_batches))
                # would otherwise overwrite them before the caller reads them
                names = list(res)
                hold = holds.get(key)
                if hold is None or hold.numel() != len(names):
                    hold = holds[key] = torch.zeros(len(names), dtype=torch.float32, device=errD.device)
                torch.stack([torch.as_tensor(res[k], dtype=torch.float32, device=errD.device).reshape(()) for k in names], out=hold)
                return {k: hold[i] for i, k in enumerate(names)}
            gc_ = calls[key] = graphs.GraphedCall(eager, "the %s critic's forward+backward" % key, bn_owner=net,
                                                  stream=self._side_stream(key), pool_from=self.__dict__.get("_cr", {}).get(key),
                                                  enabled=lambda: self._critic_graph_on(key))
        return gc_(*a)

    def _generator_forward(self, st_m, st_c, im_m, im_c, use_segment):
        """The generator pass of the G step WITH autograd (reference :367-369). Its forward and its backward are
        captured as two HIP graphs (cpcsv/graphs.GraphedAutograd): the parameter gradients land in G's flat gradient
        buffer as a side effect of the backward graph. CPCSV_G_GRAPH=0 / injected noise: eager."""
        netG = self.nets[0]
        gc_ = self.__dict__.get("_gg")
        if gc_ is None:
            def eager(a, b, c, d):
                # one stream. Tried and dropped: both halves at once like the no-grad pass (slower: this pass competes
                # with the three critic updates that already fill the GPU); halves serialised in the forward but on two
                # streams so that their backward chains overlap (-0.1 ms, and an intermittent mismatch against eager)
                import miscc.utils as MU
                if MU.BATCH_PASSES and hasattr(netG, "sample_both"):
                    (vl, st_fake, _, _, c_mu, c_logvar, _), (il, im_fake, _, _, cim_mu, cim_logvar, se_fake) = \
                        netG.sample_both(a, b, c, d, seg=use_segment)
                    return vl, st_fake, c_mu, c_logvar, il, im_fake, cim_mu, cim_logvar, se_fake
                vl, st_fake, _, _, c_mu, c_logvar, _ = netG.sample_videos(a, b)
                il, im_fake, _, _, cim_mu, cim_logvar, se_fake = netG.sample_images(c, d, seg=use_segment)
                return vl, st_fake, c_mu, c_logvar, il, im_fake, cim_mu, cim_logvar, se_fake
            gc_ = self._gg = graphs.GraphedAutograd(eager, "the generator's forward/backward", bn_owner=netG,
                                                    wgrad_stream=self._side_stream("wg") if self._streams_on() else None,
                                                    late_stream=self._late_stream(netG),
                                                    enabled=lambda: graphs.env_on("CPCSV_G_GRAPH") and netG.noise_source is None
                                                    and graphs.many_graphs_safe())
        return gc_(st_m, st_c, im_m, im_c)

    def _late_stream(self, netG):
        """Third branch of the generator's backward graph (cpcsv.runtime.set_late_stream): the decoder layers' fused optimiser
        launches wait until the data-gradient chain has left the decoder (signal: the backward of StoryGAN.fc, its first
        layer). CPCSV_LATE_UPDATES=0: they go out per layer on the weight-gradient branch as in round 2."""
        from cpcsv import modules as M
        if not (self._streams_on() and graphs.env_on("CPCSV_LATE_UPDATES")) or not isinstance(getattr(netG, "fc", None), M.FusedSequential):
            return None
        marked = 0
        for m in netG.modules():
            if isinstance(m, M.FusedSequential):
                for lay in m._plan():
                    if isinstance(lay, M.KernelLayer) and lay.fused:
                        lay.late_update = True
                        marked += 1
        first = [lay for lay in netG.fc._plan() if isinstance(lay, M.KernelLayer)]
        if not marked or not first:
            return None
        first[0].late_flush = True
        # (round 5: extra flush points at upsample1 / upsample2, so that part of the parked updates go out earlier: 13.57 / 13.59
        # against 13.45-13.48 ms per step; CPCSV_LATE_UPDATES=0, every update right behind its layer's weight gradient: 13.86)
        return self._side_stream("late")

    def _critic_score(self, key, net, a):
        """compute_generator_loss of one (frozen) critic on the new fakes (reference :386-400): forward graph + a
        backward graph that yields d loss / d fake. Captured on the critic's stream; CPCSV_SCORE_GRAPH=0: eager."""
        gpus = self.gpus
        calls = self.__dict__.setdefault("_sg", {})
        gc_ = calls.get(key)
        if gc_ is None:
            def eager(fake, real, real_labels, cate, cond):
                return compute_generator_loss(net, fake, real, real_labels, cate, cond, gpus)
            gc_ = calls[key] = graphs.GraphedAutograd(eager, "the %s critic's scoring pass" % key, bn_owner=net,
                                                      stream=self._side_stream(key), grad_inputs=(0,),
                                                      enabled=lambda: graphs.env_on("CPCSV_SCORE_GRAPH") and self._streams_on()
                                                      and self.nets[0].noise_source is None and graphs.many_graphs_safe())
        return gc_(*a)

    def _streams_on(self):
        return os.environ.get("CPCSV_STREAMS", "1") != "0"

    def _root_grad(self, loss):
        """The d loss / d loss = 1 a backward() starts from, as ONE persistent device scalar (backward() without it makes a
        ones_like - a fill launch - per call: four per step)."""
        ones = self.__dict__.setdefault("_ones", {})
        key = (loss.device, loss.dtype)
        if key not in ones:
            ones[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
        return ones[key]

    def _prepack_critic(self, key):
        """Right behind a critic's optimiser step, on that critic's stream: rebuild the operand copies of its layers whose weights
        went through the multi-tensor Adam launch (the small first conv, the logit / category heads; the big layers' copies are
        rewritten by their fused update). They are what the scoring pass and the data-gradient pass towards the fakes read;
        rebuilt lazily they sit at the head of the generator's backward chain (eight ~20 us transposing packs), here they hide
        behind the generator's forward pass. CPCSV_PREPACK=0: lazily, as before."""
        if os.environ.get("CPCSV_PREPACK", "1") == "0":
            return
        from cpcsv import modules as M
        net = {"im": self.nets[1], "st": self.nets[2], "se": self.nets[3]}[key]
        opt = self._opt_of[key]
        lays = self.__dict__.setdefault("_prepack_layers", {}).get(key)
        if lays is None:
            lays = []
            for m in net.modules():
                if isinstance(m, M.FusedSequential):
                    for lay in m._plan():
                        if isinstance(lay, M.KernelLayer):
                            lays.append(lay)
                        elif isinstance(lay, M._HeadConv):
                            lays.extend(lay._by_hw.values())
                elif type(m) is M.HeadConv2d:
                    for lay in m._layers.values():
                        if isinstance(lay, M._HeadConv):
                            lays.extend(lay._by_hw.values())
            self._prepack_layers[key] = lays
        dt = runtime.dcode()
        with torch.no_grad():
            for lay in lays:
                w = lay.holder.master()
                if w is None or opt.is_fused(w):
                    continue
                need_bwd = not getattr(lay, "logit_head", False)          # (the fused logit head only reads the forward layout)
                lay.packs(w, L_F32 if lay.compute_f32 else dt, "both" if need_bwd else "fwd")

    def _refresh_shuffle(self, b, t):
        """This step's create_random_shuffle decisions (reference miscc/utils.py:17-44; host RNGs) into the persistent device
        buffers the story critic's pass gathers through. Never inside a graph capture."""
        import miscc.utils as MU_
        buf = self.__dict__.get("_shuffle_buf")
        if buf is None or (buf.b, buf.t) != (b, t):
            buf = self._shuffle_buf = MU_.ShufflePlanBuffers(b, t, self.device)
        buf.refresh()
        return buf

    def _batch_prep(self, im_batch, st_batch, td):
        """The step's input slicing / concatenation (reference :254-264,287-288,303-304) as ONE launch (cpcsv_batch_prep), or None
        when the inputs are not plain fp32 device tensors with contiguous rows (then the torch ops do it, launch by launch)."""
        from cpcsv import kernels as K_
        idesc, ilab, icont = im_batch['description'], im_batch['labels'], im_batch['content']
        sdesc, slab = st_batch['description'], st_batch['labels']
        ts = (idesc, ilab, icont, sdesc, slab)
        if not all(t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1 for t in ts):
            return None
        if idesc.dim() != 2 or icont.dim() != 3 or sdesc.dim() != 3 or slab.dim() != 3 or ilab.dim() != 2:
            return None
        if not (ilab.is_contiguous() and slab.is_contiguous() and icont.is_contiguous() and sdesc.is_contiguous()):
            return None
        if icont.shape[0] != idesc.shape[0] or slab.shape[:2] != sdesc.shape[:2] or min(idesc.shape[1], icont.shape[2], sdesc.shape[2]) < td:
            return None
        return K_.batch_prep(idesc, ilab, icont, sdesc, slab, td)

    # ---------------------------------------------------------------- the hot path (reference :252-416)
    def train_step(self, st_batch, im_batch, next_batches=None):
        """One iteration of the reference loop body. Batches are dicts of DEVICE tensors with the keys the
        reference reads (:254-274). Returns a dict of device scalars (no host sync).

        `next_batches` is accepted for API compatibility with round 2 (a look-ahead of the critics' real-image passes that
        measured neutral and was removed: real and fake batches now go through each critic together)."""
        netG, netD_im, netD_st, netD_se = self.nets
        use_segment = cfg.SEGMENT_LEARNING and netD_se is not None
        td = cfg.TEXT.DIMENSION
        gpus = self.gpus
        # (1) batch prep, :254-288
        im_real_imgs = im_batch['images']
        im_labels = im_batch['labels']
        st_real_imgs = st_batch['images']
        st_labels = st_batch['labels']
        prep = self._batch_prep(im_batch, st_batch, td)
        if prep is not None:
            # one launch: both motion inputs (text | labels), contiguous content inputs, the per-story label presence and mean text
            im_motion_input, im_content_input, st_motion_input, st_text, st_text_mean, characters_mu = prep
        else:
            im_motion_input = torch.cat((im_batch['description'][:, :td], im_labels), 1)
            im_content_input = im_batch['content'][:, :, :td]
            st_text = st_batch['description'][:, :, :td]
            st_motion_input = torch.cat((st_text, st_labels), 2)
            st_text_mean = characters_mu = None
        st_content_input = st_text
        se_real_imgs = im_batch['images_seg'] if use_segment else None
        nim, nst = im_real_imgs.shape[0], st_real_imgs.shape[0]
        im_real_labels, im_fake_labels = self.im_real_labels[:nim], self.im_fake_labels[:nim]
        st_real_labels, st_fake_labels = self.st_real_labels[:nst], self.st_fake_labels[:nst]

        # every power iteration a critic's update will consume (2 per tower layer, 3 per head layer), on that critic's own
        # stream: it overlaps the generator pass that makes the fakes. (Enqueued BEHIND that pass's graph launch instead - the side
        # streams waiting on a step-start event - the step takes 14.82 instead of 13.61 ms: eager launches that arrive while the
        # multi-branch graph holds the hardware queues wait for it, and the iterations end up in front of the critics.)
        main = torch.cuda.current_stream()
        plans = {}
        for key, net in (("se", netD_se), ("im", netD_im), ("st", netD_st)):
            p = self._sn_plans.get(key)
            if p is None:
                continue
            if net.training:
                plans[key] = p
                side = self._side_stream(key)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    p.run("D")
            else:
                p.disarm()
        # (3a) the critics' passes over the REAL images depend on nothing the generator does: they start now, on the
        # critic streams, and overlap the generator pass below (same order per critic as the reference: real, fake)
        reals = [("im", netD_im, im_real_imgs), ("st", netD_st, st_real_imgs)]
        if use_segment:
            reals.insert(0, ("se", netD_se, se_real_imgs))
        feat_real = {}
        import miscc.utils as MU
        if MU.BATCH_PASSES:
            reals = []           # real and fake batches go through each critic TOGETHER (compute_discriminator_loss): no early pass
        for key, net, imgs in reals:
            side = self._side_stream(key)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                feat_real[key] = self._critic_real(key, net, imgs)

        # (2) fakes without grad; every module stays in train mode, :295-300
        st_fake, c_mu, im_fake, cim_mu, se_fake = self._nograd_fakes(st_motion_input, st_content_input,
                                                                      im_motion_input, im_content_input)
        # critic conditions (:303-307): the parts that do not depend on the generator are made once per step, the
        # concatenations are one launch each (no gradient flows into a condition: every consumer detaches it)
        if characters_mu is None:
            characters_mu = (st_labels.mean(1) > 0).float()                       # :303 (no host round trip)
            st_text_mean = st_text.mean(1)

        def conditions(c_mu_, cim_mu_):
            from cpcsv import kernels as K_
            st_parts = [c_mu_.detach().contiguous(), st_text_mean, characters_mu]
            im_parts = [im_motion_input, cim_mu_.detach().contiguous()]
            st_mu_ = torch.empty(nst, sum(t.shape[1] for t in st_parts), dtype=torch.float32, device=self.device)
            im_mu_ = torch.empty(nim, sum(t.shape[1] for t in im_parts), dtype=torch.float32, device=self.device)
            K_.concat_pad(st_parts, st_mu_, nst, st_mu_.shape[1])                 # :304
            K_.concat_pad(im_parts, im_mu_, nim, im_mu_.shape[1])                 # :307
            return st_mu_, im_mu_
        st_mu, im_mu = conditions(c_mu, cim_mu)

        # (3) critics, :313-346. The three critics are independent networks, so their forward/backward/Adam run
        # concurrently on three HIP streams (their small-map GEMMs fill a fraction of the 256 CUs each); results
        # are identical to the reference's order (se fwd, im fwd, st fwd; se bwd+step; im, st bwd; im, st step).
        out = {}
        jobs = []
        if use_segment:
            jobs.append(("se", netD_se, self.se_optimizerD, (se_real_imgs, se_fake, im_real_labels, im_fake_labels, im_labels, im_mu), "seg_D"))
        jobs.append(("im", netD_im, self.im_optimizerD, (im_real_imgs, im_fake, im_real_labels, im_fake_labels, im_labels, im_mu), "img_D"))
        jobs.append(("st", netD_st, self.st_optimizerD, (st_real_imgs, st_fake, st_real_labels, st_fake_labels, st_labels, st_mu), "st_D"))

        def critic_update(key, net, a, tag):
            with torch.cuda.stream(self._side_stream(key)):
                if net.seq_consisten_model is not None:         # this step's frame-shuffle decisions (host RNGs), onto the device
                    import miscc.utils as MU_
                    if torch.cuda.is_current_stream_capturing():
                        # whole-step capture: the host draws cannot live inside the graph (they would run once, at capture, and
                        # every replay would train the order critic on the same frozen plan) - train_step_graphed refreshes the
                        # device buffers before the capture and before every replay
                        buf = self.__dict__.get("_shuffle_buf")
                        if buf is None or not buf.armed or (buf.b, buf.t) != (a[0].shape[0], a[0].shape[2]):
                            raise RuntimeError("order critic inside a whole-step capture without a pre-drawn shuffle plan")
                    else:
                        buf = self._refresh_shuffle(a[0].shape[0], a[0].shape[2])
                    MU_.shuffle_buffers = buf
                    try:
                        return self._critic_backward(key, net, a, tag, feat_real.get(key))
                    finally:
                        MU_.shuffle_buffers = None
                        buf.armed = False
                return self._critic_backward(key, net, a, tag, feat_real.get(key))

        def critic_finish(key, opt):                           # collectives stay on ONE host thread, in a fixed order
            with torch.cuda.stream(self._side_stream(key)):
                self._exchange_and_step(key, opt)
                self._prepack_critic(key)
                if key in plans:
                    plans[key].run("G")      # the scoring pass's iterations, on the UPDATED weights, behind the update on this stream

        # (one host thread: a thread per critic was measured at 37 ms/step against 26 ms — the launches are short
        # enough that GIL hand-offs cost more than the overlap returns)
        for key, net, opt, a, tag in jobs:
            self._side_stream(key).wait_stream(main)
            out.update(critic_update(key, net, a, tag))
            critic_finish(key, opt)
        # The generator's own forward of step (4) reads only G's weights and fresh noise, never the critics, so it
        # is enqueued on the main stream BEFORE joining the critic streams and overlaps the whole critic update.
        # (Round 5 tried starting it even earlier, beside the no-grad pass. With that pass on a stream of its own - a FIFTH busy
        # stream, which shares a hardware queue with one of the other four - 14.18 against 13.42 ms per step; with it on the story
        # critic's stream (no fifth stream: the critic follows it there anyway) 13.46 / 13.51 against 13.48 / 13.51 - nothing:
        # the phase is throughput-bound. On the segmentation critic's stream, the first in enqueue order: 15.30.)
        # `jobs` keeps every main-stream tensor the side streams still read alive until the join below.

        # (4) generator, :365-416. Critic parameters are frozen for this pass: the reference back-props
        # into them too, but those gradients are zeroed (:313-317) before anything reads them.
        critics = [n for n in (netD_im, netD_st, netD_se) if n is not None]
        frozen = [p for n in critics for p in n.parameters() if p.requires_grad]
        try:
            self._buckets["G"].zero()      # netG.zero_grad(), reference :365
            self.optimizerG.prepare_step()
            gout = self._generator_forward(st_motion_input, st_content_input, im_motion_input, im_content_input, use_segment)
            (video_latents, st_fake, c_mu, c_logvar, image_latents, im_fake, cim_mu, cim_logvar, se_fake) = gout
            extra = None
            if video_latents is not None:                                         # cascade, :370-384
                pair = lambda lat: sum(mse_loss(g, h) for h, g in zip(lat[0], lat[1]))
                video_latent_loss = pair(video_latents)
                image_latent_loss = pair(image_latents)
                reconstruct_img = netG.train_autoencoder(se_real_imgs)
                reconstruct_fake = netG.train_autoencoder(se_fake)
                reconstruct_loss = (mse_loss(reconstruct_img, se_real_imgs) + mse_loss(reconstruct_fake, se_fake)) / 2.0
                extra = video_latent_loss + reconstruct_loss                     # :413 (image_latent_loss is logged only)
                out.update({'G/image_vae_loss': image_latent_loss.detach(), 'G/video_vae_loss': video_latent_loss.detach(),
                            'G/reconstruct_loss': reconstruct_loss.detach()})
            st_mu, im_mu = conditions(c_mu, cim_mu)
            for p in frozen:
                p.requires_grad_(False)
            for key, *_ in jobs:             # critics updated (:346) before they score the new fakes
                main.wait_stream(self._side_stream(key))
            se_errG, se_accG = 0, 0
            gjobs = [("im", netD_im, (im_fake, im_real_imgs, im_real_labels, im_labels, im_mu)),
                     ("st", netD_st, (st_fake, st_real_imgs, st_real_labels, st_labels, st_mu))]
            if use_segment:
                gjobs.insert(0, ("se", netD_se, (se_fake, se_real_imgs, im_real_labels, im_labels, im_mu)))
            gres = {}

            def critic_score(key, net, a):
                with torch.cuda.stream(self._side_stream(key)):
                    return self._critic_score(key, net, a)

            for key, net, a in gjobs:        # the critics score the fakes concurrently; autograd replays each on its stream
                self._side_stream(key).wait_stream(main)
                gres[key] = critic_score(key, net, a)
            for key, _, _ in gjobs:
                main.wait_stream(self._side_stream(key))
            if use_segment:
                se_errG, se_accG, _ = gres["se"]
            im_errG, im_accG, _ = gres["im"]
            st_errG, st_accG, st_consG = gres["st"]
            if netD_st.seq_consisten_model is not None:
                out['G/consistency'] = st_consG
            im_kl_loss = KL_loss(cim_mu, cim_logvar)                              # :402-403
            st_kl_loss = KL_loss(c_mu, c_logvar)
            # errG_total = im_errG + im_kl * KL + ratio * (se_errG * SEGMENT_RATIO + st_errG * IMAGE_RATIO + st_kl * KL)   (:409-410)
            #              [+ (video_latent_loss + reconstruct_loss) * RECONSTRUCT_LOSS, :413] - one launch each way
            terms = [(1.0, im_errG), (cfg.TRAIN.COEFF.KL, im_kl_loss), (self.ratio * cfg.IMAGE_RATIO, st_errG),
                     (self.ratio * cfg.TRAIN.COEFF.KL, st_kl_loss)]
            if use_segment:
                terms.append((self.ratio * cfg.SEGMENT_RATIO, se_errG))
            if extra is not None:
                terms.append((cfg.RECONSTRUCT_LOSS, extra))
            from cpcsv.functional import LinCombFn
            errG_total = LinCombFn.apply([float(w) for w, _ in terms], *[t for _, t in terms])
            runtime.defer_small_wgrads(True)      # (eager steps; a replayed backward graph carries its own batched launch)
            try:
                errG_total.backward(self._root_grad(errG_total))
                runtime.flush_small_wgrads()
            finally:
                runtime.defer_small_wgrads(False)
        finally:
            for p in frozen:
                p.requires_grad_(True)
        self._exchange_and_step("G", self.optimizerG)
        out.update({'G/loss': errG_total.detach(), 'G/im': im_errG.detach(), 'G/st': st_errG.detach(),
                    'G/se': se_errG.detach() if use_segment else 0.0,
                    'G/im_KL': im_kl_loss.detach(), 'G/st_KL': st_kl_loss.detach(),
                    'Accuracy/im_G': im_accG, 'Accuracy/se_G': se_accG, 'Accuracy/st_G': st_accG})
        return out

    # ---------------------------------------------------------------- whole-step HIP graph
    def train_step_graphed(self, st_batch, im_batch, warmup=3, next_batches=None):
        """train_step replayed as ONE captured HIP graph (~2000 kernel launches per step would otherwise make the
        host the bottleneck). The first `warmup` calls run eagerly (lazy buffers, weight packs, Adam tables), the next
        call captures, later calls copy the batch into the static input buffers and replay. Opt-in (CPCSV_GRAPH=1):
        with the critics on concurrent streams the eager path currently runs as fast. Everything step-dependent
        lives on the device (Adam step/lr scalars, SN u/v, BN running stats, RNG offsets via torch's graph-safe
        generator). Falls back to eager for good if capture is refused (e.g. a collective that cannot be captured)."""
        gs = self.__dict__.setdefault("_gs", {"n": 0, "graph": None, "off": os.environ.get("CPCSV_GRAPH", "0") != "1"})
        if gs["off"]:
            return self.train_step(st_batch, im_batch, next_batches)
        if "st" not in gs:
            gs["st"] = {k: v.clone() for k, v in st_batch.items() if torch.is_tensor(v)}
            gs["im"] = {k: v.clone() for k, v in im_batch.items() if torch.is_tensor(v)}
        else:
            for k, v in gs["st"].items():
                v.copy_(st_batch[k], non_blocking=True)
            for k, v in gs["im"].items():
                v.copy_(im_batch[k], non_blocking=True)
        if gs["n"] < warmup:
            gs["n"] += 1
            return self.train_step(gs["st"], gs["im"])
        netD_st = self.nets[2]
        if netD_st is not None and netD_st.seq_consisten_model is not None:
            # fresh frame-shuffle decisions for THIS step, drawn on the host outside the graph (capture and every replay alike)
            self._refresh_shuffle(gs["st"]["images"].shape[0], gs["st"]["images"].shape[2])
        if gs["graph"] is None:
            bns = [m for n in self.nets if n is not None for m in n.modules() if hasattr(m, "note_batch")]
            before = [m._pending for m in bns]
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self.train_step(gs["st"], gs["im"])
                gs["graph"], gs["out"] = g, out
                gs["bn"] = [(m, m._pending - b) for m, b in zip(bns, before)]
                for m, b in zip(bns, before):
                    m._pending = b                  # capture executes nothing; the replay below is this step
            except Exception as e:               # pragma: no cover - depends on the runtime
                gs["off"] = True
                print("[cpcsv] HIP graph capture refused (%s: %s); continuing eagerly" % (type(e).__name__, e))
                torch.cuda.synchronize()
                return self.train_step(gs["st"], gs["im"])
        gs["graph"].replay()
        for m, k in gs["bn"]:
            m._pending += k
        for opt in (self.optimizerG, self.im_optimizerD, self.st_optimizerD, self.se_optimizerD):
            if opt is not None:
                for grp in opt.param_groups:
                    grp["step"] = grp.get("step", 0) + 1
        return gs["out"]

    # ---------------------------------------------------------------- logging / epoch-end sample (reference :357-360,432-444)
    _STD_KEYS = ('st_D/loss', 'st_D/real', 'st_D/fake', 'st_D/order')
    _STD_RING = 20

    def _log_story_critic(self, stats, step):
        """The reference writes the story critic's scalars EVERY step (trainer.py:357-360; each a device->host sync there).
        Here a step parks them in a device ring (one launch, no sync) and the ring is read back with ONE copy every
        _STD_RING steps / at the end of the epoch."""
        if self.rank != 0:
            return
        ring = self.__dict__.get("_std_ring")
        if ring is None:
            ring = self._std_ring = torch.zeros(self._STD_RING, len(self._STD_KEYS), dtype=torch.float32, device=self.device)
            self._std_steps = []
        vals = [self._dev_scalar(stats.get(k, 0.0)) for k in self._STD_KEYS]
        torch.stack(vals, out=ring[len(self._std_steps)])
        self._std_steps.append(step)
        if len(self._std_steps) == self._STD_RING:
            self._flush_story_critic()

    def _dev_scalar(self, v):
        """a logged value as a 0-d fp32 device tensor WITHOUT a host->device copy per step: Python numbers (keys a configuration does
        not produce, e.g. 'st_D/order' without the order critic) come from a small cache of device constants - a pageable
        `torch.as_tensor(0.0, device=cuda)` every step blocks the host on HIP"""
        if torch.is_tensor(v):
            return v.detach().to(dtype=torch.float32).reshape(()) if v.is_cuda else v.detach().float().reshape(()).to(self.device)
        cache = self.__dict__.setdefault("_const_scalars", {})
        t = cache.get(float(v))
        if t is None:
            t = cache[float(v)] = torch.full((), float(v), dtype=torch.float32, device=self.device)
        return t

    def _flush_story_critic(self):
        steps = self.__dict__.get("_std_steps")
        if not steps:
            return
        host = self._std_ring[:len(steps)].cpu()
        for r, step in enumerate(steps):
            for c, key in enumerate(self._STD_KEYS):
                self._logger.add_scalar(key, float(host[r, c]), step)
        del steps[:]

    def _log_stats(self, stats, step):
        """The 20-step scalars (reference :432-435) with ONE device->host copy for the whole dict."""
        keys = [k for k in stats if k not in self._STD_KEYS]          # (the story critic's scalars are written every step)
        host = torch.stack([self._dev_scalar(stats[k]) for k in keys]).cpu()
        for key, v in zip(keys, host.tolist()):
            self._logger.add_scalar(key, v, step)

    def _epoch_sample(self, netG, st_batch, epoch, i):
        """reference trainer.py:437-444: at the end of every epoch the generator - still in TRAIN mode, under no_grad - renders
        the epoch's last story batch: the pass draws noise (CA eps, h0, T step noises) and advances every BatchNorm's running
        statistics like any train-mode forward, so it is part of the training trajectory, not just a dump. Every rank runs it;
        rank 0 writes the sheet."""
        use_segment = cfg.SEGMENT_LEARNING and self.nets[3] is not None
        td = cfg.TEXT.DIMENSION
        st_text = st_batch['description'][:, :, :td]
        st_motion_input = torch.cat((st_text, st_batch['labels']), 2)
        with torch.no_grad():
            _, fake, _, _, _, _, se_fake = netG.sample_videos(st_motion_input, st_text, seg=use_segment)
        if self.rank != 0 or not self.image_dir:
            return
        st_result = save_story_results(st_batch['images'].cpu(), fake, st_batch.get('text'), epoch, self.image_dir, i)
        self._logger.add_image("pororo", st_result.transpose(2, 0, 1) / 255, epoch)
        if use_segment and se_fake is not None:
            se_result = save_image_results(None, se_fake)
            self._logger.add_image("segment", se_result.transpose(2, 0, 1) / 255, epoch)

    # ---------------------------------------------------------------- epoch loop (reference :187-485)
    def train(self, imageloader, storyloader, testloader, stage=1):
        c_time = time.time()
        if cfg.EVALUATE_FID_SCORE:
            raise NotImplementedError("FID/FVD evaluation (reference trainer.py:160-174) is outside the hot path")
        self.imageloader = imageloader
        self.imagedataset = None
        netG, netD_im, netD_st, netD_se = self.setup()
        lr_decay_step = cfg.TRAIN.LR_DECAY_EPOCH
        start_epoch = int(self.con_ckpt) if self.con_ckpt else 0
        print('LR DECAY EPOCH: {}'.format(lr_decay_step))
        # host policy: the launch stream of a step is ~1900 kernel launches long and a cyclic-GC pause stalls all of
        # it, so the collector is off inside the step loop and run explicitly every 200 iterations / between epochs
        import gc
        gc.disable()
        for epoch in range(start_epoch, self.max_epoch):
            start_t = time.time()
            num_step = len(storyloader)
            gc.collect()
            def batches():                   # (story batch, image batch) in the reference's order (:250-252)
                for data in storyloader:
                    im_batch = self.sample_real_image_batch()
                    yield ingest.to_device_batch(data, self.device, feeder=True), im_batch
            feed = batches()
            cur = next(feed, None)
            last = None
            i = -1
            while cur is not None:
                i += 1
                if i % 200 == 199:
                    gc.collect()
                nxt = next(feed, None)       # one batch of look-ahead: its pinned host->device copies (cpcsv.ingest.DeviceFeeder,
                #                              own copy stream) overlap the step that is enqueued next
                ingest.wait_ready(cur[0])
                ingest.wait_ready(cur[1])
                stats = self.train_step_graphed(cur[0], cur[1], next_batches=nxt)
                last, cur = cur, nxt
                step = i + num_step * epoch
                self._log_story_critic(stats, step)                              # reference :357-360: every step
                if i % 20 == 0 and self.rank == 0:                               # reference :432-435
                    self._log_stats(stats, step)
            self._flush_story_critic()
            if last is not None:
                self._epoch_sample(netG, last[0], epoch, i)                      # reference :437-444
            # LR halving, reference :447-456 (se_optimizerD is never decayed — quirk 13)
            if epoch % lr_decay_step == 0 and epoch > 0:
                self.generator_lr *= 0.5
                for g in self.optimizerG.param_groups:
                    g['lr'] = self.generator_lr
                self.discriminator_lr *= 0.5
                for opt in (self.st_optimizerD, self.im_optimizerD):
                    for g in opt.param_groups:
                        g['lr'] = self.discriminator_lr
                for opt in (self.optimizerG, self.st_optimizerD, self.im_optimizerD):
                    opt.sync_lr()                   # device-side lr scalars: no graph re-capture needed
                lr_decay_step *= 2
            if self.rank == 0:
                self._logger.add_scalar('learning/generator', self.optimizerG.param_groups[0]['lr'], epoch)
                self._logger.add_scalar('learning/st_discriminator', self.st_optimizerD.param_groups[0]['lr'], epoch)
                self._logger.add_scalar('learning/im_discriminator', self.im_optimizerD.param_groups[0]['lr'], epoch)
                print("----[{}/{}]Epoch time:{:.1f} s, Total time:{:.2f} hours----".format(
                    epoch, self.max_epoch, time.time() - start_t, (time.time() - c_time) / 3600.0))
                if epoch % self.snapshot_interval == 0 and self.model_dir:
                    save_model(netG, netD_im, netD_st, netD_se, epoch, self.model_dir)
        gc.enable()
        if self.rank == 0 and self.model_dir:
            save_model(netG, netD_im, netD_st, netD_se, self.max_epoch, self.model_dir)
