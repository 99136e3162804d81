"""miscc/config.py — global config object with the reference's keys and merge rules.

Mirrors /root/reference/miscc/config.py (keys :9-66, strict key+type merge :68-99,
cfg_from_file :102-108) without the easydict dependency. Two reference defects are NOT reproduced
because they make the shipped yml unloadable: `yaml.load` without a Loader (fails on PyYAML>=6) is
replaced by safe_load, and a list-typed DATA_DIR placeholder is rejected with a clear message.
"""
import numpy as np


class AttrDict(dict):
    """dict with attribute access; nested dicts become AttrDict (what the reference gets from easydict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


edict = AttrDict
__C = AttrDict()
cfg = __C

__C.DATASET_NAME = 'birds'
__C.EMBEDDING_TYPE = 'cnn-rnn'
__C.CONFIG_NAME = ''
__C.GPU_ID = '0'
__C.CUDA = True
__C.WORKERS = 6
__C.VIDEO_LEN = 5
__C.NET_G = ''
__C.NET_D = ''
__C.STAGE1_G = ''
__C.DATA_DIR = ''
__C.VIS_COUNT = 64

__C.USE_SEQ_CONSISTENCY = False
__C.CONSISTENCY_RATIO = 1.0
__C.SEGMENT_LEARNING = True
__C.SEGMENT_RATIO = 1.0
__C.IMAGE_RATIO = 5.0
__C.RECONSTRUCT_LOSS = 1.0
__C.EVALUATE_FID_SCORE = False
__C.CASCADE_MODEL = True
__C.Z_DIM = 100
__C.IMSIZE = 64
__C.SESIZE = 64
__C.STAGE = 1
__C.LABEL_NUM = 9

__C.TRAIN = AttrDict()
__C.TRAIN.FLAG = True
__C.TRAIN.IM_BATCH_SIZE = 64
__C.TRAIN.ST_BATCH_SIZE = 64
__C.TRAIN.MAX_EPOCH = 600
__C.TRAIN.SNAPSHOT_INTERVAL = 50
__C.TRAIN.PRETRAINED_MODEL = ''
__C.TRAIN.PRETRAINED_EPOCH = 600
__C.TRAIN.LR_DECAY_EPOCH = 600
__C.TRAIN.DISCRIMINATOR_LR = 2e-4
__C.TRAIN.GENERATOR_LR = 2e-4
__C.TRAIN.SEGMENT_NAME = 'img_segment'
__C.TRAIN.COEFF = AttrDict()
__C.TRAIN.COEFF.KL = 2.0

__C.GAN = AttrDict()
__C.GAN.CONDITION_DIM = 124
__C.GAN.Z_DIM = 100
__C.GAN.DF_DIM = 124
__C.GAN.GF_DIM = 256
__C.GAN.GF_SEG_DIM = 1024
__C.GAN.R_NUM = 4

__C.TEXT = AttrDict()
__C.TEXT.DIMENSION = 356


def _merge_a_into_b(a, b):
    """Clobber options of b with those of a; unknown keys and type changes are errors
    (reference miscc/config.py:68-99)."""
    if not isinstance(a, AttrDict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old_type = type(b[k])
        if old_type is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(b[k]), type(v), k))
        if isinstance(v, AttrDict):
            try:
                _merge_a_into_b(a[k], b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a yml file and merge it into the defaults (reference miscc/config.py:102-108)."""
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = AttrDict(yaml.safe_load(f))
    _merge_a_into_b(yaml_cfg, __C)


def cfg_from_dict(d):
    _merge_a_into_b(AttrDict(d), __C)
