"""GPU parity of the image-prior encoder (SURVEY.md §8f "next" #1; reference depth_model.py:110-118: timm
tf_efficientnetv2_s, features_only) -- TF-"SAME" padded convs with SiLU epilogue, depthwise 3x3 + squeeze-excite,
and the whole 40-block pyramid -- against the CPU oracle.  Parity unpinned against timm itself (absent): the oracle
restates the published architecture and is cross-checked against an ATen restatement in test_oracle_effnet.py."""
import numpy as np
import pytest
import torch
from torch import nn

import oracle
from parity import assert_close
from simplerecon_amd import _lib, ops, synthetic
from simplerecon_amd import image_encoder
from simplerecon_amd.image_encoder import EfficientNetV2SFeatures

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def _bn(c, seed):
    return synthetic.seeded_fill_(nn.BatchNorm2d(c, eps=1e-3), seed=seed).eval()


@pytest.mark.parametrize("shape,cout,k,stride", [
    ((2, 3, 64, 96), 24, 3, 2),      # stem: 3 input channels, even map -> pads (0, 0, 1, 1)
    ((1, 24, 37, 51), 96, 3, 2),     # odd map -> pads (1, 1, 1, 1)
    ((2, 48, 18, 22), 192, 3, 1),    # stride 1: symmetric
    ((1, 24, 36, 1), 24, 3, 2),      # one column
    ((1, 64, 9, 11), 256, 1, 1),     # 1x1 expansion
    ((3, 160, 5, 6), 960, 1, 1),
])
def test_same_conv_bn_silu(shape, cout, k, stride):
    rng = np.random.default_rng(sum(shape) + cout)
    x = rng.standard_normal(shape, dtype=np.float32)
    conv = synthetic.seeded_fill_(nn.Conv2d(shape[1], cout, k, stride=stride, padding=k // 2, bias=False),
                                  seed=cout).to(DEV)
    bn = _bn(cout, seed=k).to(DEV)
    with torch.inference_mode():
        y = ops.conv2d(torch.from_numpy(x).to(DEV), conv, bn=bn, act="silu", tf_same=True)
    sd = {k_: _np(v) for k_, v in bn.state_dict().items()}
    ref = oracle.silu(oracle.batchnorm_eval(oracle.conv2d_same(x, _np(conv.weight), stride), sd, "", eps=1e-3))
    assert tuple(y.shape) == ref.shape
    assert_close(y, ref, what=f"SAME conv{k}x{k}/s{stride}+bn+silu {shape}->{cout}")


def test_silu_epilogue_with_residual_winograd():
    """act="silu" on a Winograd-eligible shape (SiLU epilogue of sr_wino_kernel)."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 64, 60, 80), dtype=np.float32)
    r = rng.standard_normal((2, 64, 60, 80), dtype=np.float32)
    conv = synthetic.seeded_fill_(nn.Conv2d(64, 64, 3, padding=1, bias=True), seed=2).to(DEV)
    with torch.inference_mode():
        y = ops.conv2d(torch.from_numpy(x).to(DEV), conv, residual=torch.from_numpy(r).to(DEV), act="silu")
    ref = oracle.silu(oracle.conv2d(x, _np(conv.weight), _np(conv.bias), residual=r))
    assert_close(y, ref, what="conv3x3 + residual + silu")
    with pytest.raises(ValueError), torch.inference_mode():
        ops.conv2d(torch.from_numpy(x).to(DEV), conv, leaky=0.2, act="silu")


@pytest.mark.parametrize("shape,stride", [((2, 256, 30, 40), 2), ((1, 512, 15, 20), 1), ((2, 768, 9, 11), 1),
                                          ((1, 960, 30, 40), 2), ((3, 1536, 5, 3), 1), ((1, 8, 7, 33), 2),
                                          ((1, 72, 3, 3), 1)])
def test_dwconv_se_scale(shape, stride):
    b, c, h, w = shape
    rng = np.random.default_rng(sum(shape) + stride)
    x = rng.standard_normal(shape, dtype=np.float32)
    conv = synthetic.seeded_fill_(nn.Conv2d(c, c, 3, stride=stride, padding=1, groups=c, bias=False), seed=c).to(DEV)
    bn = _bn(c, seed=stride).to(DEV)
    rd = max(1, c // 24)
    se_r = synthetic.seeded_fill_(nn.Conv2d(c, rd, 1), seed=5).to(DEV)
    se_e = synthetic.seeded_fill_(nn.Conv2d(rd, c, 1), seed=6).to(DEV)
    with torch.inference_mode():
        d, pool = ops.dwconv3x3(torch.from_numpy(x).to(DEV), conv, bn=bn, act="silu", tf_same=True, want_pool=True)
        d0 = d.clone()
        gate = ops.se_gate(pool, d.shape[2] * d.shape[3], se_r, se_e)
        d1, gate1 = ops.se_scale_(d0.clone(), pool, se_r, se_e, want_gate=True)
        ops.scale_channels_(d, gate)
    sd = {k_: _np(v) for k_, v in bn.state_dict().items()}
    ref = oracle.silu(oracle.batchnorm_eval(oracle.dwconv3x3_same(x, _np(conv.weight), stride), sd, "", eps=1e-3))
    assert tuple(d0.shape) == ref.shape
    assert_close(d0, ref, what=f"depthwise conv+bn+silu {shape}/s{stride}")
    mean = ref.mean(axis=(2, 3))
    assert_close(pool.sum(1) / (ref.shape[2] * ref.shape[3]), mean, tol=1e-5, what="squeeze-excite average pool")
    hid = oracle.silu(mean @ _np(se_r.weight)[:, :, 0, 0].T + _np(se_r.bias))
    g = 1.0 / (1.0 + np.exp(-(hid @ _np(se_e.weight)[:, :, 0, 0].T + _np(se_e.bias))))
    assert_close(gate, g, tol=1e-5, what="squeeze-excite gate")
    assert_close(d, ref * g[:, :, None, None], what="gated depthwise output")
    assert_close(gate1, g, tol=1e-5, what="squeeze-excite gate (two-launch path)")
    assert_close(d1, ref * g[:, :, None, None], what="gated depthwise output (two-launch path)")


@pytest.mark.parametrize("shape,mid", [((8, 128, 30, 40), 512), ((2, 160, 30, 40), 960), ((8, 256, 15, 20), 1536),
                                       ((1, 128, 7, 9), 512), ((3, 160, 16, 20), 960)])
def test_mbconv_front_half_in_one_launch(shape, mid):
    """sr_mbconv_expand_dw_se_fwd (csrc/sr_mbconv_fused.hip, r05): 1x1 expansion + BN + SiLU -> depthwise 3x3 + BN + SiLU ->
    squeeze-excite pool -> gates in ONE launch, against the oracle's restatement of the same operators and against the r04
    launch-per-operator path; deterministic, and the arrival counters come back zeroed (the call can be repeated)."""
    b, ci, h, w = shape
    rng = np.random.default_rng(sum(shape) + mid)
    x = rng.standard_normal(shape, dtype=np.float32)
    pw = synthetic.seeded_fill_(nn.Conv2d(ci, mid, 1, bias=False), seed=ci).to(DEV)
    bn1 = _bn(mid, seed=1).to(DEV)
    dw = synthetic.seeded_fill_(nn.Conv2d(mid, mid, 3, padding=1, groups=mid, bias=False), seed=mid).to(DEV)
    bn2 = _bn(mid, seed=2).to(DEV)
    se = image_encoder.SqueezeExcite(mid, ci // 4)
    synthetic.seeded_fill_(se, seed=7)
    se = se.to(DEV)
    xt = torch.from_numpy(x).to(DEV)
    assert ops.mbconv_fused_supported(xt, pw, dw, se)
    with torch.inference_mode():
        d, gate = ops.mbconv_expand_dw_se(xt, pw, bn1, dw, bn2, se)
        d2, gate2 = ops.mbconv_expand_dw_se(xt, pw, bn1, dw, bn2, se)
        t = ops.conv2d(xt, pw, bn=bn1, act="silu")
        d_ref, pool = ops.dwconv3x3(t, dw, bn=bn2, act="silu", tf_same=True, want_pool=True)
        gate_ref = ops.se_gates(pool, h * w, se.conv_reduce, se.conv_expand)
    torch.cuda.synchronize()
    assert torch.equal(d, d2) and torch.equal(gate, gate2)
    cnt = ops._MBX_COUNTERS[(xt.device, torch.cuda.current_stream(xt.device).cuda_stream)]   # (one buffer per stream since r06)
    assert int(cnt.abs().sum()) == 0
    sd1 = {k_: _np(v) for k_, v in bn1.state_dict().items()}
    sd2 = {k_: _np(v) for k_, v in bn2.state_dict().items()}
    e = oracle.silu(oracle.batchnorm_eval(oracle.conv2d(x, _np(pw.weight), None), sd1, "", eps=1e-3))
    ref = oracle.silu(oracle.batchnorm_eval(oracle.dwconv3x3_same(e, _np(dw.weight), 1), sd2, "", eps=1e-3))
    assert_close(d, ref, what=f"fused expansion + depthwise {shape} -> {mid}")
    assert_close(d, d_ref, tol=1e-5, what="fused vs launch-per-operator path")
    mean = ref.mean(axis=(2, 3))
    hid = oracle.silu(mean @ _np(se.conv_reduce.weight)[:, :, 0, 0].T + _np(se.conv_reduce.bias))
    g = 1.0 / (1.0 + np.exp(-(hid @ _np(se.conv_expand.weight)[:, :, 0, 0].T + _np(se.conv_expand.bias))))
    assert_close(gate, g, tol=1e-5, what="squeeze-excite gates")
    assert_close(gate, gate_ref, tol=1e-5, what="gates vs the two-launch path")


@pytest.mark.parametrize("shape,cout,res", [((8, 1536, 15, 20), 256, True), ((1, 1536, 15, 20), 256, False),
                                            ((2, 960, 30, 40), 160, True), ((1, 512, 5, 7), 128, False),
                                            ((3, 768, 9, 11), 160, True)])
def test_deep_projection_conv_splitk(shape, cout, res):
    """1x1 projections with a long channel chain on few pixels: the split-K plan of sr_conv2d_splitk_nhwc_fwd
    (partials + deterministic finish) against the oracle, with folded BatchNorm and the block's identity skip."""
    b, c, h, w = shape
    rng = np.random.default_rng(c + cout)
    x = rng.standard_normal(shape, dtype=np.float32)
    r = rng.standard_normal((b, cout, h, w), dtype=np.float32) if res else None
    conv = synthetic.seeded_fill_(nn.Conv2d(c, cout, 1, bias=False), seed=cout).to(DEV)
    bn = _bn(cout, seed=7).to(DEV)
    xt = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last)
    rt = torch.from_numpy(r).to(DEV).contiguous(memory_format=torch.channels_last) if res else None
    with torch.inference_mode():
        y = ops.conv2d(xt, conv, bn=bn, residual=rt)
        y2 = ops.conv2d(xt, conv, bn=bn, residual=rt)
    sd = {k_: _np(v) for k_, v in bn.state_dict().items()}
    ref = oracle.batchnorm_eval(oracle.conv2d(x, _np(conv.weight)), sd, "", eps=1e-3)
    if res:
        ref = ref + r
    assert torch.equal(y, y2), "split-K finish must be deterministic"
    assert_close(y, ref, what=f"1x1 projection {shape}->{cout}")


def test_add_inplace_on_channel_slice():
    rng = np.random.default_rng(3)
    buf = torch.from_numpy(rng.standard_normal((2, 17, 19, 48), dtype=np.float32)).to(DEV).permute(0, 3, 1, 2)
    y = buf[:, 8:32]
    x = torch.from_numpy(rng.standard_normal((2, 24, 17, 19), dtype=np.float32)).to(DEV)
    ref = _np(y) + _np(x)
    keep = _np(buf).copy()
    with torch.inference_mode():
        ops.add_(y, x)
    assert np.array_equal(_np(y), ref)
    after = _np(buf)
    assert np.array_equal(after[:, :8], keep[:, :8]) and np.array_equal(after[:, 32:], keep[:, 32:])


def _encoder(seed):
    return synthetic.seeded_fill_(EfficientNetV2SFeatures(), seed=seed, gain=1.0).to(DEV)


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 72, 88), (1, 3, 480, 640)])
def test_efficientnetv2_s_pyramid_vs_oracle(shape):
    enc = _encoder(seed=3)
    rng = np.random.default_rng(shape[2])
    img = rng.standard_normal(shape, dtype=np.float32)
    with torch.inference_mode():
        feats = enc(torch.from_numpy(img).to(DEV))
    torch.cuda.synchronize()
    sd = {k: _np(v) for k, v in enc.state_dict().items()}
    ref = oracle.efficientnetv2_s_features(img, sd)
    assert [f.shape[1] for f in feats] == enc.num_ch_enc == [24, 48, 64, 160, 256]
    for i, (f, r) in enumerate(zip(feats, ref)):
        assert tuple(f.shape) == r.shape, (i, f.shape, r.shape)
        assert_close(f, r, what=f"image-prior feature {i} (stride {2 << i}) for {shape}")


def test_batch_independence_and_determinism():
    enc = _encoder(seed=4)
    g = torch.Generator().manual_seed(1)
    img = torch.randn((3, 3, 96, 128), generator=g).to(DEV)
    with torch.inference_mode():
        a = [f.clone() for f in enc(img)]
        b = enc(img)
        one = enc(img[1:2])
    for fa, fb, f1 in zip(a, b, one):
        assert torch.equal(fa, fb)
        assert_close(fa[1:2], f1, tol=1e-5, what="batch independence")


def test_refuses_training_mode_and_cpu():
    enc = EfficientNetV2SFeatures()
    with pytest.raises(_lib.HipLibraryError):
        with torch.inference_mode():
            enc(torch.zeros(1, 3, 64, 64))
    # training mode = the differentiable graph with batch-statistics BatchNorm (tests/test_gpu_encoder_training.py)
    enc = enc.to(DEV)
    enc.train()
    feats = enc(torch.randn(2, 3, 64, 64, device=DEV))
    assert all(f.requires_grad for f in feats) and [f.shape[1] for f in feats] == [24, 48, 64, 160, 256]


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 73, 89), (3, 480, 640), (1, 2, 2)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_rgb_stem_kernel(shape, channels_last):
    """sr_rgb_stem3x3s2_fwd (conv_stem 3 -> 24, stride 2, TF-"SAME", folded BatchNorm, SiLU as a VALU kernel) against ATen in
    float64 and against the implicit-GEMM path it replaces; NCHW and channels-last images, even and odd sizes."""
    import torch.nn.functional as F
    from simplerecon_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 3 + W)
    conv = torch.nn.Conv2d(3, 24, 3, stride=2, padding=1, bias=False).to(DEV)
    bn = synthetic.seeded_fill_(torch.nn.BatchNorm2d(24, eps=1e-3).eval(), seed=5).to(DEV)
    img = torch.randn((B, 3, H, W), generator=g).to(DEV)
    if channels_last:
        img = img.contiguous(memory_format=torch.channels_last)
    pt, pl, pb, pr = ops.tf_same_pads(H, W, 3, 2)
    scale, shift = ops.bn_affine(bn)
    ref = F.conv2d(F.pad(img.double(), (pl, pr, pt, pb)), conv.weight.double(), None, stride=2)
    ref = F.silu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    with torch.inference_mode():
        got = ops.rgb_stem3x3s2(img, conv, bn=bn, act="silu", tf_same=True)
        old = ops.conv2d(img, conv, bn=bn, act="silu", tf_same=True)
    assert tuple(got.shape) == tuple(ref.shape) == (B, 24, -(-H // 2), -(-W // 2))
    assert_close(got, ref.float(), tol=1e-5, what="rgb stem vs ATen")
    assert_close(got, old, tol=1e-5, what="rgb stem vs implicit GEMM")
