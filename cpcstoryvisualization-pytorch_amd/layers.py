"""layers.py — the per-sample 1-D dynamic filter, one HIP launch per call.

Drop-in for the reference's layers.py. Only `DynamicFilterLayer1D` is live in the reference
(model.py:243 imports it under the alias DynamicFilterLayer); the 2-D `DynamicFilterLayer`
(layers.py:10-59) is never instantiated and cannot run, so it is not provided.
"""
import torch.nn as nn

from cpcsv import functional as F


class DynamicFilterLayer1D(nn.Module):
    """out[n,0,x] = sum_c sum_k image[n,c,x+k-pad] * filters[n,0,c,k]  (reference layers.py:62-80).

    The reference loops N conv1d launches and concatenates; here it is one kernel (and one for the
    backward), block per sample with the 3x124 signal and 3x21 taps staged in LDS."""

    def __init__(self, filter_size, stride=1, pad=0):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("the reference only ever uses stride 1 (model.py:310-311)")
        self.filter_size, self.stride, self.pad = filter_size, stride, pad

    def forward(self, _input, **kwargs):
        image, filters = _input[0], _input[1]
        n = image.shape[0]
        taps = filters.reshape(n, image.shape[1], filters.shape[-1])
        return F.DynFilter1dFn.apply(image, taps, self.pad)
