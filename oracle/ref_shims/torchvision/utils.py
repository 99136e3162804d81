def make_grid(*a, **k):
    raise NotImplementedError("torchvision stub (oracle/ref_shims)")


def save_image(*a, **k):
    raise NotImplementedError("torchvision stub (oracle/ref_shims)")
