"""Minimal stand-in for the `easydict` package (absent from this image).

Only used by oracle/gen_golden.py to import the reference's miscc/config.py
(/root/reference/miscc/config.py:6). Nested dicts are converted to the same
class because the reference's _merge_a_into_b tests `type(a) is edict`
(miscc/config.py:72,92). Test infrastructure only; never imported by the product.
"""


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        src = dict(d or {})
        src.update(kw)
        for k, v in src.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        dict.__setitem__(self, k, v)

    __setitem__ = __setattr__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e
