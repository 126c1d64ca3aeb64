"""GPU parity of the matching-feature encoder (reference modules/networks.py:149-205) --
stem conv7x7+BN+ReLU, MaxPool+BlurPool, layer1, InstanceNorm tail -- against the CPU oracle and the golden
vectors of the reference's ResnetMatchingEncoder."""
import numpy as np
import pytest
import torch
from torch import nn

import golden_cases as gc
import oracle
from parity import assert_close
from simplerecon_amd import ops, synthetic
from simplerecon_amd.networks import ResnetMatchingEncoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nchw(t):
    return t.detach().cpu().contiguous().numpy()


@pytest.mark.parametrize("name", list(gc.MATCHING_CASES))
def test_matching_encoder_stages_vs_golden(name):
    case = gc.MATCHING_CASES[name]
    enc = synthetic.seeded_fill_(ResnetMatchingEncoder(18, 16), seed=case["seed"]).to(DEV).eval()
    gold = gc.load_golden("matching", name)
    x = gc.matching_input(case).to(DEV)
    net = enc.net
    with torch.inference_mode():
        stem = ops.stem7x7(x, net[0], net[1])
        pool = ops.maxblurpool(stem)
        t = pool
        for blk in net[4]:
            u = ops.conv2d(t, blk.conv1, bn=blk.bn1, leaky=0.0)
            t = ops.conv2d(u, blk.conv2, bn=blk.bn2, residual=t, leaky=0.0)
        out = enc(x)
    torch.cuda.synchronize()
    assert_close(stem, gold["stem"], what=f"{name} stem conv+bn+relu")
    assert_close(pool, gold["pool"], what=f"{name} maxpool+blurpool")
    assert_close(t, gold["layer1"], what=f"{name} layer1")
    assert tuple(out.shape) == gold["out"].shape
    assert_close(out, gold["out"], what=f"{name} matching features vs reference golden")
    sd = {k: v.cpu().numpy() for k, v in enc.state_dict().items()}
    assert_close(out, oracle.resnet_matching_encoder(_nchw(x), sd), what=f"{name} matching features vs oracle")


def test_stem_channels_last_image_and_no_bn():
    """The stem reads the image through its strides (NCHW or channels-last) and without a BatchNorm."""
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 3, 50, 70), dtype=np.float32)).to(DEV)
    conv = synthetic.seeded_fill_(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=True), seed=3).to(DEV)
    with torch.inference_mode():
        a = ops.stem7x7(x, conv, None, leaky=None)
        b = ops.stem7x7(x.contiguous(memory_format=torch.channels_last), conv, None, leaky=None)
    assert torch.equal(a, b)
    ref = oracle.conv2d(_nchw(x), _nchw(conv.weight), _nchw(conv.bias), stride=2, pad=3)
    assert_close(a, ref, what="stem conv (bias, no BN, no activation)")


@pytest.mark.parametrize("shape", [(1, 8, 9, 11), (2, 64, 16, 24), (1, 4, 4, 4), (1, 12, 31, 6)])
def test_maxblurpool(shape):
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape, dtype=np.float32)
    with torch.inference_mode():
        y = ops.maxblurpool(torch.from_numpy(x).to(DEV))
    ref = oracle.blurpool4_s2(oracle.maxpool2_s1(x))
    assert tuple(y.shape) == ref.shape
    assert_close(y, ref, tol=1e-6, what=f"maxblurpool {shape}")


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (1, 8, 17, 23), (3, 64, 48, 64), (2, 16, 37, 18), (1, 64, 240, 320), (2, 4, 19, 16)])
def test_maxblurpool_streaming_form_equals_the_block_kernel(shape, sr_option):
    """r05: the row-streaming kernel (2 output columns x 4 channels per thread walking down a band of rows; frame pixels in
    trailing workgroups of the same launch) against the oracle and, bit for bit, against the block kernel it replaces --
    odd / even sizes (one-output column blocks, clamped loads), several bands, writes into / reads from channel slices."""
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape, dtype=np.float32)
    xd = torch.from_numpy(x).to(DEV)
    with torch.inference_mode():
        sr_option("SR_POOL_STREAM", 1)
        y = ops.maxblurpool(xd)
        buf_in = ops.empty_nhwc(shape[0], shape[1] + 8, shape[2], shape[3], DEV).normal_()
        buf_in[:, 4:4 + shape[1]] = xd
        ho, wo = y.shape[2], y.shape[3]
        buf_out = ops.empty_nhwc(shape[0], shape[1] + 12, ho, wo, DEV).fill_(5.0)
        ops.maxblurpool(buf_in[:, 4:4 + shape[1]], out=buf_out[:, 8:8 + shape[1]])
        sr_option("SR_POOL_STREAM", 0)
        y0 = ops.maxblurpool(xd)
    torch.cuda.synchronize()
    assert_close(y, oracle.blurpool4_s2(oracle.maxpool2_s1(x)), tol=1e-6, what=f"streaming maxblurpool {shape}")
    assert torch.equal(y, y0), f"{int((y != y0).sum())} elements differ from the block kernel"
    assert torch.equal(buf_out[:, 8:8 + shape[1]], y)
    assert bool((buf_out[:, :8] == 5).all()) and bool((buf_out[:, 8 + shape[1]:] == 5).all())


@pytest.mark.parametrize("shape,leaky", [((2, 128, 10, 18), 0.2), ((3, 16, 30, 40), None), ((1, 16, 120, 160), None),
                                         ((1, 48, 7, 5), 0.2), ((2, 128, 120, 160), 0.2)])
def test_instance_norm(shape, leaky):
    rng = np.random.default_rng(sum(shape))
    # channel-dependent mean and spread, as a conv output has
    x = rng.standard_normal(shape, dtype=np.float32) * rng.uniform(0.2, 3.0, size=(1, shape[1], 1, 1)).astype(np.float32) \
        + rng.uniform(-2, 2, size=(shape[0], shape[1], 1, 1)).astype(np.float32)
    xt = torch.from_numpy(x).to(DEV)
    with torch.inference_mode():
        y = ops.instance_norm(xt, leaky=leaky)
        y2 = ops.instance_norm(xt.clone(memory_format=torch.channels_last), leaky=leaky, inplace=True)
    assert torch.equal(y, y2), "in-place and out-of-place results differ"
    ref = oracle.instance_norm(x.astype(np.float64), leaky=leaky)
    assert_close(y, ref, tol=2e-5, what=f"instance_norm {shape}")
    # deterministic (no atomics): a second launch is bit-identical
    with torch.inference_mode():
        assert torch.equal(y, ops.instance_norm(xt, leaky=leaky))


def test_replicate_padded_conv():
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 128, 10, 18), dtype=np.float32)
    conv = synthetic.seeded_fill_(nn.Conv2d(128, 16, 3, padding=1, padding_mode="replicate"), seed=4).to(DEV)
    with torch.inference_mode():
        y = ops.conv2d(torch.from_numpy(x).to(DEV), conv)
    ref = oracle.conv2d_replicate(x, _nchw(conv.weight), _nchw(conv.bias))
    assert_close(y, ref, tol=1e-5, what="replicate-padded conv3x3")
    zero = oracle.conv2d(x, _nchw(conv.weight), _nchw(conv.bias))
    assert np.abs(_nchw(y)[:, :, 1:-1, 1:-1] - zero[:, :, 1:-1, 1:-1]).max() < 1e-4  # interior = zero-padded conv


@pytest.mark.parametrize("mode,with_norm", [("replicate", True), ("zeros", True), ("replicate", False)])
def test_conv3x3_c16_with_fused_input_norm(mode, with_norm):
    """sr_conv3x3_c16_nhwc_fwd = conv3x3(LeakyReLU(InstanceNorm(x))) without materialising the normalised tensor."""
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((2, 64, 11, 19), dtype=np.float32) * 1.7 + 0.4).astype(np.float32)
    conv = synthetic.seeded_fill_(nn.Conv2d(64, 12, 3, padding=1, padding_mode=mode), seed=8).to(DEV)
    xt = torch.from_numpy(x).to(DEV)
    with torch.inference_mode():
        stats = ops.instance_norm_stats(xt) if with_norm else None
        y = ops.conv3x3_c16(xt, conv, in_stats=stats, in_leaky=0.2 if with_norm else None)
    xin = oracle.instance_norm(x, leaky=0.2) if with_norm else x
    wgt, bias = _nchw(conv.weight), _nchw(conv.bias)
    ref = oracle.conv2d_replicate(xin, wgt, bias) if mode == "replicate" else oracle.conv2d(xin, wgt, bias)
    assert tuple(y.shape) == ref.shape
    assert_close(y, ref, tol=1e-5, what=f"conv3x3_c16 {mode} norm={with_norm}")
    if with_norm:
        m = x.astype(np.float64).mean(axis=(2, 3))
        assert_close(stats[:, 0], m, tol=1e-5, what="instance_norm_stats mean")


@pytest.mark.parametrize("shape,bias", [((2, 64, 11, 19), True), ((1, 64, 8, 8), False), ((3, 64, 120, 160), True),
                                        ((1, 64, 1, 3), True)])
def test_conv1x1_with_instance_norm_statistics(shape, bias):
    """sr_conv1x1_stats_nhwc_fwd = Conv2d(64, 128, 1) and the InstanceNorm statistics of its output in one pass
    (ragged last 64-pixel tile, a tile-sized image, no bias)."""
    rng = np.random.default_rng(sum(shape))
    x = (rng.standard_normal(shape, dtype=np.float32) * 1.3 + 0.2).astype(np.float32)
    conv = synthetic.seeded_fill_(nn.Conv2d(64, 128, 1, bias=bias), seed=6).to(DEV)
    xt = torch.from_numpy(x).to(DEV)
    with torch.inference_mode():
        y, stats = ops.conv1x1_stats(xt, conv, eps=1e-5)
        y2, stats2 = ops.conv1x1_stats(xt, conv, eps=1e-5)
        two_pass = ops.instance_norm_stats(y, eps=1e-5)
    assert torch.equal(y, y2) and torch.equal(stats, stats2), "not deterministic"
    ref = oracle.conv2d(x, _nchw(conv.weight), _nchw(conv.bias) if bias else None, pad=0)
    assert_close(y, ref, tol=1e-5, what=f"conv1x1_stats output {shape}")
    r64 = ref.astype(np.float64)
    assert_close(stats[:, 0], r64.mean(axis=(2, 3)), tol=1e-5, what="fused mean")
    assert_close(stats[:, 1], 1.0 / np.sqrt(r64.var(axis=(2, 3)) + 1e-5), tol=2e-5, what="fused rstd")
    assert_close(stats, two_pass, tol=2e-5, what="fused vs two-pass statistics")
    with pytest.raises(Exception, match="Conv2d\\(64, 128, 1\\)"), torch.inference_mode():
        ops.conv1x1_stats(xt, nn.Conv2d(64, 64, 1).to(DEV))


def test_batchnorm_fold_matches_unfolded_oracle():
    rng = np.random.default_rng(12)
    x = rng.standard_normal((1, 64, 12, 20), dtype=np.float32)
    conv = synthetic.seeded_fill_(nn.Conv2d(64, 64, 3, padding=1, bias=False), seed=6).to(DEV)
    bn = synthetic.seeded_fill_(nn.BatchNorm2d(64), seed=7).to(DEV).eval()
    with torch.inference_mode():
        y = ops.conv2d(torch.from_numpy(x).to(DEV), conv, bn=bn, leaky=0.0)
        bn.running_mean.add_(0.5)  # the packed-weight cache follows parameter / buffer updates
        y_shift = ops.conv2d(torch.from_numpy(x).to(DEV), conv, bn=bn, leaky=0.0)
    sd = {k: _nchw(v) for k, v in bn.state_dict().items()}
    sd0 = dict(sd, running_mean=sd["running_mean"] - 0.5)
    ref = np.maximum(oracle.batchnorm_eval(oracle.conv2d(x, _nchw(conv.weight)), sd0, ""), 0)
    assert_close(y, ref, tol=1e-5, what="conv + folded BatchNorm + ReLU")
    ref2 = np.maximum(oracle.batchnorm_eval(oracle.conv2d(x, _nchw(conv.weight)), sd, ""), 0)
    assert_close(y_shift, ref2, tol=1e-5, what="conv + folded BatchNorm after a buffer update")


def test_full_size_properties(monkeypatch):
    """640x480 images (BASELINE cfg2/3 size): InstanceNorm'd outputs have zero mean / unit variance per image and
    channel, and every image is processed independently of its batch neighbours -- bitwise while the launch plan is the same;
    since r05 the 3x3 layers take F(4x4) or F(2x2) Winograd by how well the batch fills the chip (csrc/sr_wino4.hip,
    sr_conv_prefers_wino4: 3 images = 240 work items -> F(4x4), 1 image = 80 -> F(2x2)), and then to the fp32 bar."""
    enc = synthetic.seeded_fill_(ResnetMatchingEncoder(18, 16), seed=1).to(DEV).eval()
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((3, 3, 480, 640), generator=g).to(DEV)
    with torch.inference_mode():
        y = enc(x)
        y1 = enc(x[1:2])
    assert tuple(y.shape) == (3, 16, 120, 160)
    assert_close(y1, y[1:2], tol=1e-5, what="image 1 alone vs inside the batch (F(2x2) vs F(4x4) layer1)")
    with monkeypatch.context() as mp:
        mp.setattr(ops, "WINO4_MODE", 0)     # the same kernel at both batch sizes: bit for bit
        with torch.inference_mode():
            z, z1 = enc(x), enc(x[1:2])
    assert torch.equal(z[1:2], z1)
    assert_close(y, z, tol=1e-5, what="F(4x4) vs F(2x2) layer1")
    m = y.double().mean(dim=(2, 3))
    v = y.double().var(dim=(2, 3), unbiased=False)
    assert float(m.abs().max()) < 1e-4 and float((v - 1).abs().max()) < 1e-3, (float(m.abs().max()), float((v - 1).abs().max()))


def test_forward_pair_equals_concatenated_forward():
    """forward_pair (no image concatenation) = forward on cat([cur, src]) regrouped, bit for bit."""
    enc = synthetic.seeded_fill_(ResnetMatchingEncoder(18, 16), seed=2).to(DEV).eval()
    g = torch.Generator(device="cpu").manual_seed(4)
    cur = torch.randn((2, 3, 64, 96), generator=g).to(DEV)
    src = torch.randn((2, 3, 3, 64, 96), generator=g).to(DEV)
    with torch.inference_mode():
        c, s_ = enc.forward_pair(cur, src)
        ref = enc(torch.cat([cur.unsqueeze(1), src], dim=1).flatten(0, 1)).unflatten(0, (2, 4))
    assert tuple(c.shape) == (2, 16, 16, 24) and tuple(s_.shape) == (2, 3, 16, 16, 24)
    assert torch.equal(c, ref[:, 0]) and torch.equal(s_, ref[:, 1:])


def test_training_mode_and_bad_configs_fail_loudly():
    enc = ResnetMatchingEncoder(18, 16).to(DEV)
    y = enc.train()(torch.randn(2, 3, 32, 32, device=DEV))   # training mode: the differentiable graph, batch-statistics BN
    assert y.requires_grad and tuple(y.shape) == (2, 16, 8, 8)
    with pytest.raises(ValueError):
        ResnetMatchingEncoder(17, 16)
    with pytest.raises(ValueError), torch.inference_mode():
        ops.stem7x7(torch.zeros(1, 4, 32, 32, device=DEV), enc.net[0], enc.net[1])
    assert enc.eval()(torch.randn(1, 3, 32, 32, device=DEV)).requires_grad   # eval-mode BatchNorm, gradients still flow
    with torch.no_grad():
        assert not enc(torch.randn(1, 3, 32, 32, device=DEV)).requires_grad   # the fused inference kernels
