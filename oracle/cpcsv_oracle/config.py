"""Config values the hot path reads (reference: miscc/config.py:9-66, cfg/final.yml).

The reference keeps these in a global mutable EasyDict read at module construction
(model.py:217-222,490-493). The oracle passes an explicit immutable object instead.
"""
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class OracleCfg:
    video_len: int = 5          # VIDEO_LEN            cfg/final.yml:12
    label_num: int = 9          # LABEL_NUM            cfg/final.yml:14
    text_dim: int = 356         # TEXT.DIMENSION       cfg/final.yml:41-42
    cond_dim: int = 124         # GAN.CONDITION_DIM    cfg/final.yml:36
    z_dim: int = 100            # GAN.Z_DIM            miscc/config.py:60
    df_dim: int = 124           # GAN.DF_DIM           cfg/final.yml:37
    gf_dim: int = 256           # GAN.GF_DIM (x8 inside G, model.py:218)
    gf_seg_dim: int = 1024      # GAN.GF_SEG_DIM       cfg/final.yml:39
    segment_learning: bool = True   # SEGMENT_LEARNING cfg/final.yml:19
    segment_ratio: float = 1.0      # SEGMENT_RATIO
    image_ratio: float = 5.0        # IMAGE_RATIO
    reconstruct_loss: float = 1.0   # RECONSTRUCT_LOSS  miscc/config.py:31
    cascade: bool = False           # CASCADE_MODEL     cfg/final.yml:17
    use_seq_consistency: bool = False   # USE_SEQ_CONSISTENCY  cfg/final.yml:16, miscc/config.py:26
    consistency_ratio: float = 1.0      # CONSISTENCY_RATIO    cfg/final.yml:18
    kl_coeff: float = 1.0           # TRAIN.COEFF.KL    cfg/final.yml:33
    g_lr: float = 1e-4              # TRAIN.GENERATOR_LR
    d_lr: float = 4e-4              # TRAIN.DISCRIMINATOR_LR
    st_batch: int = 12              # BASELINE config 2 (trainer.py:263 comment)
    im_batch: int = 60
    # hard-coded in the reference generator (model.py:226-230)
    dfl_channels: int = 3
    dfl_taps: int = 21
    dfl_width: int = 124

    @property
    def motion_dim(self):  # model.py:220
        return self.text_dim + self.label_num

    @property
    def ngf(self):  # model.py:218
        return self.gf_dim * 8

    @property
    def joint_dim(self):  # model.py:244
        return self.motion_dim + self.cond_dim + self.dfl_width

    @property
    def critic_cond_dim(self):  # model.py:516
        return self.cond_dim + self.text_dim + self.label_num

    def but(self, **kw):
        return replace(self, **kw)


def pororo_cfg(**kw):
    """cfg/final.yml dims with the ST=12/IM=60 batch BASELINE.json config 2 names."""
    return OracleCfg(**kw)


def tiny_cfg(**kw):
    """Same code path, narrow widths: small enough to commit goldens."""
    base = dict(gf_dim=8, gf_seg_dim=32, df_dim=8, text_dim=20, label_num=3,
                st_batch=3, im_batch=4)
    base.update(kw)
    return OracleCfg(**base)


def clevr_cfg(**kw):
    """CLEVR-shaped dims inferred from datasets/clevr.py:24,38-41,104 (cfg/clevr.yml is
    not shipped): T=4, text 72, labels 15. batch=1 is impossible (BatchNorm1d), use 2/8."""
    base = dict(video_len=4, text_dim=72, label_num=15, st_batch=2, im_batch=8)
    base.update(kw)
    return OracleCfg(**base)
