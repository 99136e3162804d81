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

# The reference's option table (miscc/config.py:9-66), kept as ONE nested literal: key names, nesting and value TYPES are the
# contract (the yml merge below rejects unknown keys and type changes), the values are the reference's defaults.
_DEFAULTS = {
    "DATASET_NAME": "birds", "EMBEDDING_TYPE": "cnn-rnn", "CONFIG_NAME": "", "GPU_ID": "0", "CUDA": True, "WORKERS": 6,
    "VIDEO_LEN": 5, "NET_G": "", "NET_D": "", "STAGE1_G": "", "DATA_DIR": "", "VIS_COUNT": 64,
    "USE_SEQ_CONSISTENCY": False, "CONSISTENCY_RATIO": 1.0, "SEGMENT_LEARNING": True, "SEGMENT_RATIO": 1.0,
    "IMAGE_RATIO": 5.0, "RECONSTRUCT_LOSS": 1.0, "EVALUATE_FID_SCORE": False, "CASCADE_MODEL": True,
    "Z_DIM": 100, "IMSIZE": 64, "SESIZE": 64, "STAGE": 1, "LABEL_NUM": 9,
    "TRAIN": {"FLAG": True, "IM_BATCH_SIZE": 64, "ST_BATCH_SIZE": 64, "MAX_EPOCH": 600, "SNAPSHOT_INTERVAL": 50,
              "PRETRAINED_MODEL": "", "PRETRAINED_EPOCH": 600, "LR_DECAY_EPOCH": 600, "DISCRIMINATOR_LR": 2e-4,
              "GENERATOR_LR": 2e-4, "SEGMENT_NAME": "img_segment", "COEFF": {"KL": 2.0}},
    "GAN": {"CONDITION_DIM": 124, "Z_DIM": 100, "DF_DIM": 124, "GF_DIM": 256, "GF_SEG_DIM": 1024, "R_NUM": 4},
    "TEXT": {"DIMENSION": 356},
}
cfg = AttrDict(_DEFAULTS)
__C = cfg


def _merge_a_into_b(a, b, _where=""):
    """Overwrite the options of `b` with those of `a` (reference miscc/config.py:68-99): a key `b` does not have is a
    KeyError, a value whose type differs from the default's is a ValueError (numpy defaults take the default's dtype)."""
    if not isinstance(a, AttrDict):
        return
    for key in a:
        new, here = a[key], _where + key
        if key not in b:
            raise KeyError('{} is not a valid config key'.format(key))
        cur = b[key]
        if type(cur) is not type(new):
            if not isinstance(cur, np.ndarray):
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(cur), type(new), key))
            new = np.array(new, dtype=cur.dtype)
        if isinstance(new, AttrDict):
            try:
                _merge_a_into_b(new, cur, here + ".")
            except (KeyError, ValueError):
                print('Error under config key: {}'.format(key))
                raise
        else:
            b[key] = new


def cfg_from_file(filename):
    """Load a yml file and merge it into the defaults (reference miscc/config.py:102-108)."""
    import yaml
    with open(filename, 'r') as f:
        loaded = yaml.safe_load(f)
    _merge_a_into_b(AttrDict(loaded), cfg)
