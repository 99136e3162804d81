"""GPU parity of the whole training step: product (HIP, via the C ABI) vs the oracle, on the golden
fixtures generated from the real reference (same weights, batch and noise)."""
import pytest
import torch

from tests import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_step_fp32_matches_oracle(tag):
    """fp32 mode (exact f32 MFMA). Tolerances: losses 2e-4 rel; each net's whole gradient vector
    within 5e-3 in relative L2, and every element within 5e-2 of its tensor's max (a BN output within round-off of 0 may flip one LeakyReLU mask and move
    one element of a small-sample sum; SURVEY §8(c): BN + spectral norm amplify round-off)."""
    pu.run_step_parity(tag, "fp32")


def test_step_fp32_two_stream_nograd_pass():
    """Same parity with the no-grad generator pass split into its story half and image half on two HIP streams
    (what every step after the first does): per-branch descriptors, ordered BatchNorm running-stat updates."""
    pu.run_step_parity("plain", "fp32", two_stream=True)


@pytest.mark.parametrize("tag", ["plain", "cascade"])
def test_step_bf16_within_band(tag):
    """bf16 operands / fp32 accumulate at the fixture's TINY widths (2-32 channels: the harshest case for
    bf16): losses within 5 % after one step. At cfg/final.yml widths bf16 tracks fp32 within 1 % on every
    loss over consecutive steps (tools/gpu_diag.py bf16full; profiles/r01_bf16_vs_fp32.txt)."""
    pu.run_step_parity(tag, "bf16")


def test_no_native_fallback_is_loaded():
    """The process must have the in-tree HIP library mapped; nothing else provides the ops."""
    from cpcsv import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libcpcsv_hip.so" in maps
