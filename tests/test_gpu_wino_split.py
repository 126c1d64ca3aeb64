"""The split-precision variant of the Winograd convolution (SR_WINO_SPLIT=bf16|f16: the multiply on the 16-bit matrix pipe,
two 16-bit pieces per fp32 operand, three products; a fenced experiment) through the SAME checks, at the
SAME tolerances, as the fp32-MFMA kernel: the Winograd tests of tests/test_gpu_conv.py re-run under the switch."""
import numpy as np
import pytest
import torch

import test_gpu_conv as tc
from parity import rel_err
from simplerecon_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = ["bf16", "f16"]


@pytest.fixture(params=MODES)
def split(request, monkeypatch, sr_option):
    sr_option("SR_WINO_SPLIT", request.param)
    return request.param


def _param_values(fn):
    return [m for m in fn.pytestmark if m.name == "parametrize"][0].args[1]


@pytest.mark.parametrize("shape", _param_values(tc.test_winograd_split_k))
def test_split_k_plans(shape, split):
    tc.test_winograd_split_k(shape)


@pytest.mark.parametrize("shape", _param_values(tc.test_winograd_pipeline_cases_and_work_order))
def test_structural_cases_and_work_order(shape, split, monkeypatch, sr_option):
    tc.test_winograd_pipeline_cases_and_work_order(shape, monkeypatch, sr_option)


@pytest.mark.parametrize("shape", _param_values(tc.test_winograd_many_regions_per_workgroup_with_a_channel_tail))
def test_border_epilogue_with_channel_tail(shape, split):
    tc.test_winograd_many_regions_per_workgroup_with_a_channel_tail(shape)


def test_dispatcher_sweep(split):
    """The randomised dispatcher sweep: layers the split kernel does not cover (unaligned tensors, channel counts that are no
    multiple of 4) take the direct fp32 kernel; everything stays inside the sweep's tolerance."""
    tc.test_conv_random_shapes_and_options()


def test_basic_blocks_and_encoder_decoder(split):
    for name in _param_values(tc.test_basic_block):
        tc.test_basic_block(name)
    for name in _param_values(tc.test_encoder_decoder):
        tc.test_encoder_decoder(name)


@pytest.mark.parametrize("shape", [(8, 64, 240, 320, 64), (8, 192, 120, 160, 64), (2, 384, 15, 20, 384)])
def test_error_against_fp64_next_to_the_fp32_kernel(shape, monkeypatch, sr_option):
    """f16 pieces (weights pre-scaled by 2^8): as close to an fp64 convolution as the fp32-MFMA kernel (measured: 0.9-1.0 x its
    rms error); bf16 pieces: 16-18 bits, ~25 x the fp32 kernel's rms error -- still 1e-6 of the output range."""
    B, ci, H, W, co = shape
    monkeypatch.setattr(ops, "WINO4_MODE", 0)   # "the fp32 kernel" = F(2x2) sr_wino_kernel, the one the split kernels restate
    g = torch.Generator().manual_seed(ci + H)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
    x = torch.randn((B, ci, H, W), generator=g).to(DEV)
    res = torch.randn((B, co, H, W), generator=g).to(DEV)
    with torch.no_grad():
        ref = torch.nn.functional.leaky_relu(
            torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double(), 0.2)
    rms = {}
    for mode in ("0", "bf16", "f16"):
        sr_option("SR_WINO_SPLIT", mode)
        with torch.inference_mode():
            y = ops.conv2d(x, conv, residual=res, leaky=0.2)
        rms[mode] = float(((y.double() - ref) ** 2).mean().sqrt() / ref.abs().max())
    print(shape, rms)
    assert rms["f16"] < 1.5 * rms["0"]
    assert rms["bf16"] < 2e-6 and rms["0"] < 1e-7


def test_unknown_mode_fails_loudly(monkeypatch, sr_option):
    from simplerecon_amd._lib import HipLibraryError
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV)
    x = torch.randn((1, 64, 32, 32), device=DEV)
    sr_option("SR_WINO_SPLIT", "int8")
    with pytest.raises(HipLibraryError), torch.inference_mode():
        ops.conv2d(x, conv)


def test_f16_pieces_fail_loudly_outside_fp16_range(monkeypatch, sr_option):
    """|V| >= 65504 has no fp16 high piece: the f16 variant returns non-finite values there (never a silently saturated
    one); the bf16 variant, which has fp32's exponent range, computes the layer."""
    g = torch.Generator().manual_seed(3)
    conv = torch.nn.Conv2d(16, 32, 3, padding=1).to(DEV)
    x = (torch.randn((1, 16, 24, 32), generator=g) * 1e5).to(DEV)
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    with torch.inference_mode():
        sr_option("SR_WINO_SPLIT", "f16")
        y = ops.conv2d(x, conv)
        assert not bool(torch.isfinite(y).all())
        sr_option("SR_WINO_SPLIT", "bf16")
        y = ops.conv2d(x, conv)
        assert bool(torch.isfinite(y).all()) and rel_err(y, ref.float()) < 2e-5
