"""GPU parity of the whole training step: product (HIP, via the C ABI) vs the oracle, on the golden
fixtures generated from the real reference (same weights, batch and noise)."""
import pytest
import torch

from tests import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_step_fp32_matches_oracle(tag):
    """fp32 mode (exact f32 MFMA). Tolerances: losses 2e-4 rel, gradients 5e-3 of the per-tensor max
    (BN + spectral norm amplify round-off)."""
    pu.run_step_parity(tag, "fp32")


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_step_bf16_within_band(tag):
    """bf16 operands / fp32 accumulate: losses within 3 % after one step (north_star asks 1 % on the
    1k-step loss curve; a single tiny-width step is the harsher case)."""
    pu.run_step_parity(tag, "bf16")


def test_no_native_fallback_is_loaded():
    """The process must have the in-tree HIP library mapped; nothing else provides the ops."""
    from cpcsv import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libcpcsv_hip.so" in maps
