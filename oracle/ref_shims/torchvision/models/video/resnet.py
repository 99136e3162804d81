def r2plus1d_18(*a, **k):
    raise NotImplementedError("torchvision stub (oracle/ref_shims)")
