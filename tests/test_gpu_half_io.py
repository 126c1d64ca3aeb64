"""16-bit kernel I/O under torch.autocast (VERDICT r03 "what's missing" #1; reference options.py:100-101 `precision: 16`,
train.py:132): the Winograd and the pointwise convolution kernels read fp16 / bf16 activations (four channels = one 8-byte
load), accumulate in fp32 and round once on the way out.

* kernel level: the 16-bit-I/O entry points equal the fp32 entry points run on the widened inputs, rounded to the I/O dtype --
  BIT FOR BIT (same arithmetic, one extra rounding);
* operator level: inside an autocast region the conv stack's activations and saved tensors are 16-bit, the outputs agree with
  the fp32 run to half precision, every gradient exists, is finite and agrees with the fp32 gradients to half precision;
  SR_AUTOCAST_HALF_IO=0 restores the fp32 kernels."""
import ctypes as C

import numpy as np
import pytest
import torch

from parity import rel_err
from simplerecon_amd import _lib, autograd_ops, ops, synthetic
from simplerecon_amd.networks import CVEncoder, DepthDecoderPP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]


def _wino_io(x, conv, res, slope, out):
    lib = _lib.lib()
    b, ci, h, w = x.shape
    co = conv.out_channels
    wp = torch.empty(lib.sr_wino_packed_weight_floats(co, ci), dtype=torch.float32, device=DEV)
    st = _lib.stream_ptr(x.device)
    _lib.check(lib.sr_wino_pack_weights(_lib.ptr(conv.weight.detach().contiguous()), co, ci, _lib.ptr(wp), st), "pack")
    isb, isp = ops._strides(x)
    osb, osp = ops._strides(out)
    rsb, rsp = ops._strides(res) if res is not None else (0, 0)
    io = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[x.dtype]
    rc = lib.sr_conv3x3_wino_io_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(conv.bias.detach()), _lib.ptr(res), rsb, rsp,
                                         _lib.ptr(out), osb, osp, b, h, w, ci, co, C.c_float(-1.0 if slope is None else slope), io, st)
    _lib.check(rc, "sr_conv3x3_wino_io_nhwc_fwd")
    return out


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 64, 48, 64, 64, True, 0.2), (1, 24, 40, 56, 24, False, None), (3, 32, 17, 23, 96, True, 0.0),
                                   (1, 128, 240, 320, 64, True, 0.2), (2, 64, 16, 16, 32, False, 0.2)])
def test_winograd_16bit_io_equals_the_rounded_fp32_kernel(shape, dt):
    B, ci, H, W, co, with_res, slope = shape
    g = torch.Generator().manual_seed(ci + co + H)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
    wide = torch.randn((B, ci + 8, H, W), generator=g).to(DEV).to(dt).contiguous(memory_format=torch.channels_last)
    x = wide[:, 4:4 + ci]                                    # a channel slice: 8-byte aligned rows
    res = torch.randn((B, co, H, W), generator=g).to(DEV).to(dt).contiguous(memory_format=torch.channels_last) if with_res else None
    out = torch.empty((B, co + 4, H, W), dtype=dt, device=DEV).contiguous(memory_format=torch.channels_last).fill_(7.0)
    with torch.inference_mode():
        _wino_io(x, conv, res, slope, out[:, 4:])
        ref = torch.empty((B, co, H, W), dtype=torch.float32, device=DEV).contiguous(memory_format=torch.channels_last)
        _wino_io(x.float().contiguous(memory_format=torch.channels_last), conv,
                 res.float() if res is not None else None, slope, ref)
    assert bool((out[:, :4] == 7).all())
    assert torch.equal(out[:, 4:], ref.to(dt)), float((out[:, 4:].float() - ref).abs().max())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 192, 30, 40, 64, True, None), (1, 64, 33, 47, 1, False, None), (3, 112, 12, 20, 100, True, 0.2),
                                   (2, 256, 15, 20, 128, False, 0.0)])
def test_pointwise_16bit_io_equals_the_rounded_fp32_kernel(shape, dt):
    B, ci, H, W, co, with_res, slope = shape
    lib = _lib.lib()
    g = torch.Generator().manual_seed(ci * 3 + co)
    conv = torch.nn.Conv2d(ci, co, 1).to(DEV)
    x = torch.randn((B, ci, H, W), generator=g).to(DEV).to(dt).contiguous(memory_format=torch.channels_last)
    res = torch.randn((B, co, H, W), generator=g).to(DEV).to(dt).contiguous(memory_format=torch.channels_last) if with_res else None
    wp = torch.empty(lib.sr_conv_packed_weight_floats(co, ci, 1), dtype=torch.float32, device=DEV)
    st = _lib.stream_ptr(x.device)
    _lib.check(lib.sr_conv_pack_weights(_lib.ptr(conv.weight.detach().contiguous()), co, ci, 1, _lib.ptr(wp), st), "pack")
    sl = C.c_float(-1.0 if slope is None else slope)

    def run(xx, rr, io):
        out = torch.empty((B, co, H, W), dtype=xx.dtype, device=DEV).contiguous(memory_format=torch.channels_last)
        isb, isp = ops._strides(xx)
        osb, osp = ops._strides(out)
        rsb, rsp = ops._strides(rr) if rr is not None else (0, 0)
        _lib.check(lib.sr_pw_conv_io_nhwc_fwd(_lib.ptr(xx), isb, isp, _lib.ptr(wp), _lib.ptr(conv.bias.detach()), _lib.ptr(rr), rsb,
                                              rsp, _lib.ptr(out), osb, osp, B, H * W, ci, co, sl, io, st), "sr_pw_conv_io_nhwc_fwd")
        return out
    with torch.inference_mode():
        got = run(x, res, {torch.float16: 1, torch.bfloat16: 2}[dt])
        ref = run(x.float(), res.float() if res is not None else None, 0)
    assert torch.equal(got, ref.to(dt)), float((got.float() - ref).abs().max())


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_stack_trains_with_16bit_activations_under_autocast(dt, monkeypatch):
    """CVEncoder -> DepthDecoderPP under torch.autocast: activations between the layers and what autograd saves are 16-bit,
    the log-depth outputs agree with the fp32 run to half precision and so do the parameter gradients."""
    monkeypatch.setattr(autograd_ops, "HALF_IO", True)       # (opt-in since r06: SR_AUTOCAST_HALF_IO=1 SR_AUTOCAST_HALF_STORAGE=1)
    monkeypatch.setattr(autograd_ops, "STORE_HALF", True)
    torch.manual_seed(0)
    enc = synthetic.seeded_fill_(CVEncoder(16, [8, 12, 16, 24], [16, 24, 32, 48]), seed=1).to(DEV)
    dec = synthetic.seeded_fill_(DepthDecoderPP([6] + enc.num_ch_enc), seed=2).to(DEV)
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 32, 48
    vol = torch.randn((B, 16, H, W), generator=g).to(DEV).requires_grad_(True)
    pyr = [torch.randn((B, c, H >> i, W >> i), generator=g).to(DEV) for i, c in enumerate([8, 12, 16, 24])]
    f0 = torch.randn((B, 6, 2 * H, 2 * W), generator=g).to(DEV)

    def step(autocast):
        for m in (enc, dec):
            m.zero_grad(set_to_none=True)
        vol.grad = None
        saved = {"bytes": 0, "half": 0, "n": 0}

        def pack(t):
            if t.dim() == 4 and not isinstance(t, torch.nn.Parameter):
                saved["bytes"] += t.numel() * t.element_size()
                saved["n"] += 1
                saved["half"] += int(t.dtype == dt)
            return t
        with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
            with torch.autocast("cuda", dtype=dt, enabled=autocast):
                out = dec([f0] + enc(vol, pyr))
                loss = sum(out[f"log_depth_pred_s{i}_b1hw"].float().abs().mean() for i in range(4))
        loss.backward()
        grads = {n: p.grad.clone() for m in (enc, dec) for n, p in m.named_parameters() if p.grad is not None}
        return out, grads, vol.grad.clone(), saved

    ref_out, ref_g, ref_dv, s32 = step(False)
    out, gr, dv, s16 = step(True)
    assert out["log_depth_pred_s0_b1hw"].dtype == dt and s16["half"] >= 0.9 * s16["n"], s16
    assert s16["bytes"] < 0.6 * s32["bytes"], (s16, s32)
    tol = 3e-2 if dt == torch.float16 else 2e-1          # 8 mantissa bits for bf16, ~20 layers deep
    for i in range(4):
        k = f"log_depth_pred_s{i}_b1hw"
        assert torch.isfinite(out[k]).all() and rel_err(out[k].float(), ref_out[k]) < tol, (k, rel_err(out[k].float(), ref_out[k]))
    assert sorted(gr) == sorted(ref_g) and dv.dtype == torch.float32 and torch.isfinite(dv).all()

    def l2(a, b):
        return float((a.float() - b).norm() / b.norm().clamp_min(1e-12))
    errs = [l2(gr[n], ref_g[n]) for n in gr]
    assert all(torch.isfinite(v).all() for v in gr.values())
    assert np.median(errs) < (3e-2 if dt == torch.float16 else 2e-1), (np.median(errs), max(errs))
    # the switch: fp32 kernels again, the r03 behaviour (identical forward, 16-bit storage only)
    monkeypatch.setattr(autograd_ops, "HALF_IO", False)
    out0, _, _, _ = step(True)
    assert out0["log_depth_pred_s0_b1hw"].dtype == torch.float32
    assert torch.equal(out0["log_depth_pred_s0_b1hw"], ref_out["log_depth_pred_s0_b1hw"])


def test_half_inputs_outside_autocast_are_refused():
    conv = torch.nn.Conv2d(16, 16, 3, padding=1).to(DEV)
    x = torch.randn((1, 16, 8, 8), device=DEV).half().requires_grad_(True)
    with pytest.raises((TypeError, _lib.HipLibraryError)):
        autograd_ops.conv_bias_act(x, conv, slope=0.2)
