"""GPU parity of the HIP conv stack (BasicBlock, upsample, CVEncoder, DepthDecoderPP) against
the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle
from parity import assert_close, rel_err
from simplerecon_amd import ops, synthetic
from simplerecon_amd.layers import BasicBlock
from simplerecon_amd.networks import CVEncoder, DepthDecoderPP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", list(gc.BLOCK_CASES))
def test_basic_block(name):
    case = gc.BLOCK_CASES[name]
    blk = synthetic.seeded_fill_(BasicBlock(case["cin"], case["cout"], stride=case["stride"]), seed=case["seed"]).to(DEV)
    x = gc.block_input(case)
    with torch.inference_mode():
        y = blk(x.to(DEV))
    torch.cuda.synchronize()
    gold = gc.load_golden("block", name)["out"]
    sd = {k: v.cpu().numpy() for k, v in blk.state_dict().items()}
    y_o = oracle.basic_block(x.numpy(), sd, "", stride=case["stride"])
    assert tuple(y.shape) == gold.shape
    assert_close(y, y_o, tol=1e-5, what=f"BasicBlock {name} vs oracle")
    assert_close(y, gold, what=f"BasicBlock {name} vs reference golden")


def test_basic_block_writes_into_concat_slice():
    case = gc.BLOCK_CASES["same"]
    blk = synthetic.seeded_fill_(BasicBlock(16, 16), seed=case["seed"]).to(DEV)
    x = gc.block_input(case).to(DEV)
    with torch.inference_mode():
        ref = blk(x)
        buf = ops.empty_nhwc(x.shape[0], 40, x.shape[2], x.shape[3], DEV).fill_(7.0)
        blk(x, out=buf[:, 8:24])
        # reading from a slice as well
        again = blk(buf[:, 8:24] * 1.0)
        again2 = blk(buf[:, 8:24])
    assert torch.equal(buf[:, 8:24], ref) and bool((buf[:, :8] == 7).all()) and bool((buf[:, 24:] == 7).all())
    assert torch.equal(again, again2)


def test_upsample():
    x = gc.upsample_input()
    gold = np.load(gc.GOLDEN_DIR + "/upsample.npz")["out"]
    with torch.inference_mode():
        y = ops.upsample2x(x.to(DEV))
    assert_close(y, gold, tol=1e-6, what="upsample2x")
    assert_close(y, oracle.upsample2x(x.numpy()), tol=1e-6, what="upsample2x vs oracle")


@pytest.mark.parametrize("shape", [(1, 8, 1, 1), (2, 64, 7, 5), (1, 16, 1, 9), (3, 12, 6, 1), (2, 64, 30, 40), (1, 256, 15, 20)])
def test_upsample_block_form_equals_the_per_pixel_kernel(shape, sr_option):
    """r05: one thread per 2 x 2 output block (4 loads for 4 stores) against ATen's bilinear x2 (align_corners=False) and, bit
    for bit, against the per-pixel kernel: 1-pixel maps, odd sizes, a concat-slice destination, non-finite inputs (the
    clamped border multiplies its second row / column by 0 in both kernels, as ATen does)."""
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    b, c, h, w = shape
    with torch.inference_mode():
        sr_option("SR_UPSAMPLE_QUAD", 1)
        y = ops.upsample2x(x)
        buf = ops.empty_nhwc(b, c + 8, 2 * h, 2 * w, DEV).fill_(3.0)
        ops.upsample2x(x, out=buf[:, 4:4 + c])
        xn = x.clone()
        xn[0, 0, h - 1, w - 1] = float("inf")
        xn[0, -1, 0, 0] = float("nan")
        yn = ops.upsample2x(xn)
        sr_option("SR_UPSAMPLE_QUAD", 0)
        y0, yn0 = ops.upsample2x(x), ops.upsample2x(xn)
    ref = torch.nn.functional.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=False)
    assert_close(y, ref, tol=1e-6, what=f"upsample2x {shape}")
    assert torch.equal(y, y0)
    assert torch.equal(buf[:, 4:4 + c], y) and bool((buf[:, :4] == 3).all()) and bool((buf[:, 4 + c:] == 3).all())
    assert torch.equal(torch.isnan(yn), torch.isnan(yn0)) and torch.equal(yn.nan_to_num(7.0, 8.0, 9.0), yn0.nan_to_num(7.0, 8.0, 9.0))


@pytest.mark.parametrize("name", list(gc.NET_CASES))
def test_encoder_decoder(name):
    case = gc.NET_CASES[name]
    enc = synthetic.seeded_fill_(CVEncoder(case["D"], case["enc_ch"][1:], case["cv_outs"]), seed=case["seed"]).to(DEV)
    dec = synthetic.seeded_fill_(DepthDecoderPP(case["enc_ch"][:1] + case["cv_outs"]), seed=case["seed"] + 1).to(DEV)
    vol, feats = gc.net_inputs(case)
    gold = gc.load_golden("net", name)
    with torch.inference_mode():
        cvf = enc(vol.to(DEV), [f.to(DEV) for f in feats[1:]])
        outs = dec([feats[0].to(DEV)] + cvf)
    torch.cuda.synchronize()
    for i, t in enumerate(cvf):
        assert_close(t, gold[f"cv_feat_{i}"], what=f"{name} CVEncoder level {i}")
    assert list(outs) == [f"log_depth_pred_s{i}_b1hw" for i in (3, 2, 1, 0)]
    for k, v in outs.items():
        assert tuple(v.shape) == gold[k].shape
        assert_close(v, gold[k], what=f"{name} decoder {k}")


def test_conv_properties_at_full_resolution():
    """240x320 (the decoder's dominant level at 640x480): linearity of a bias-free conv and
    agreement of a strided-row subset with the oracle."""
    torch.manual_seed(0)
    blk = synthetic.seeded_fill_(BasicBlock(64, 64), seed=3).to(DEV)
    x = torch.randn(1, 64, 240, 320)
    with torch.inference_mode():
        y = blk(x.to(DEV))
        conv = blk.conv1
        a = ops.conv2d(x.to(DEV), conv)
        b = ops.conv2d((2 * x).to(DEV), conv)
    bias = conv.bias.view(1, -1, 1, 1)
    assert rel_err(b - bias, 2 * (a - bias)) < 1e-6
    # oracle on a crop that includes the image border (receptive field of the block = 5x5)
    crop = x[:, :, :24, :40].contiguous()
    sd = {k: v.cpu().numpy() for k, v in blk.state_dict().items()}
    y_o = oracle.basic_block(crop.numpy(), sd, "")
    assert_close(y[:, :, :20, :36], y_o[:, :, :20, :36], tol=1e-5, what="240x320 block crop vs oracle")


@pytest.mark.parametrize("shape", [(1, 256, 15, 20, 128), (2, 384, 15, 20, 64), (1, 256, 30, 40, 256), (1, 320, 15, 20, 64),
                                   (1, 192, 8, 12, 32)])
def test_winograd_split_k(shape):
    """Deep low-resolution layers run split-K (partial outputs + deterministic reduce): same result as the oracle,
    bias / residual / LeakyReLU applied once, bit-reproducible."""
    from simplerecon_amd import _lib
    b, ci, h, w, co = shape
    assert _lib.lib().sr_wino_splitk_factor(b, h, w, ci, co) > 1, "the launch plan should split this shape"
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((b, ci, h, w), dtype=np.float32)
    res = rng.standard_normal((b, co, h, w), dtype=np.float32)
    conv = synthetic.seeded_fill_(torch.nn.Conv2d(ci, co, 3, padding=1), seed=5).to(DEV)
    xt, rt = torch.from_numpy(x).to(DEV), torch.from_numpy(res).to(DEV)
    with torch.inference_mode():
        y = ops.conv2d(xt, conv, residual=rt, leaky=0.2)
        y2 = ops.conv2d(xt, conv, residual=rt, leaky=0.2)
    assert torch.equal(y, y2)
    ref = oracle.conv2d(x, conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy(), residual=res, leaky=0.2)
    assert_close(y, ref, tol=1e-5, what=f"split-K Winograd conv {shape}")


@pytest.mark.parametrize("shape", [(1, 256, 30, 40, 384, False), (1, 128, 60, 80, 256, False), (2, 192, 17, 23, 96, True),
                                   (1, 144, 9, 7, 40, True)])
def test_strided_conv_split_k(shape, sr_option):
    """The 3x3 / stride-2 down-convolutions of CVEncoder (reference modules/networks.py:38-55) on small maps run split-K
    since r06 (sr_conv2d_splitk_nhwc_fwd: partial outputs + the deterministic finish that applies bias / residual / LeakyReLU
    once): the oracle's result, bit-reproducible, and -- where the plan really splits -- the low-order bits of a different
    summation order than the unsplit launch."""
    from simplerecon_amd import _lib
    b, ci, h, w, co, with_res = shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    rng = np.random.default_rng(sum(shape[:5]))
    x = rng.standard_normal((b, ci, h, w), dtype=np.float32)
    res = rng.standard_normal((b, co, ho, wo), dtype=np.float32) if with_res else None
    conv = synthetic.seeded_fill_(torch.nn.Conv2d(ci, co, 3, stride=2, padding=1), seed=11).to(DEV)
    xt = torch.from_numpy(x).to(DEV)
    rt = torch.from_numpy(res).to(DEV) if with_res else None
    assert _lib.lib().sr_conv_splitk_workspace_bytes(b, h, w, ci, co, 3, 2) > 0, "this shape should be offered a split-K workspace"
    with torch.inference_mode():
        y = ops.conv2d(xt, conv, residual=rt, leaky=0.2)
        y2 = ops.conv2d(xt, conv, residual=rt, leaky=0.2)
        sr_option("SR_CONV_KSPLIT", 0)
        y1 = ops.conv2d(xt, conv, residual=rt, leaky=0.2)
    assert torch.equal(y, y2), "split-K finish must be deterministic"
    ref = oracle.conv2d(x, conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy(), stride=2, residual=res, leaky=0.2)
    assert_close(y, ref, tol=1e-5, what=f"split-K strided conv {shape}")
    assert_close(y1, ref, tol=1e-5, what=f"unsplit strided conv {shape}")
    assert_close(y, y1, tol=5e-6, what="split vs unsplit plan (summation order over K = 9 Cin only)")


def test_conv_random_shapes_and_options():
    """Randomised sweep over the conv dispatcher (direct / Winograd / split-K, 32- or 64-channel blocks, vector or
    scalar epilogue): odd sizes, channel counts that are not multiples of 4 / 16 / 32, with and without bias,
    residual, activation, writing into a channel slice -- each against the C oracle."""
    rng = np.random.default_rng(2024)
    for case in range(28):
        k = int(rng.choice([1, 3, 3, 3]))
        stride = int(rng.choice([1, 1, 1, 2])) if k == 3 else 1
        ci = int(rng.choice([3, 8, 16, 24, 30, 64, 100, 128, 200, 320]))
        co = int(rng.choice([1, 5, 16, 32, 48, 64, 96, 130]))
        b = int(rng.choice([1, 2, 3]))
        h, w = int(rng.integers(5, 34)), int(rng.integers(5, 50))
        use_bias, use_res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        leaky = [None, 0.2, 0.0][int(rng.integers(0, 3))]
        conv = synthetic.seeded_fill_(torch.nn.Conv2d(ci, co, k, stride=stride, padding=k // 2, bias=use_bias),
                                      seed=case).to(DEV)
        x = rng.standard_normal((b, ci, h, w), dtype=np.float32)
        ho, wo = ops.conv_out_hw(h, w, stride, k)
        res = rng.standard_normal((b, co, ho, wo), dtype=np.float32) if use_res else None
        xt = torch.from_numpy(x).to(DEV)
        if case % 3 == 0:
            xt = xt.contiguous(memory_format=torch.channels_last)
        rt = torch.from_numpy(res).to(DEV) if use_res else None
        with torch.inference_mode():
            if case % 4 == 1:   # into a channel slice of a wider buffer
                buf = ops.empty_nhwc(b, co + 7, ho, wo, DEV).fill_(3.0)
                y = ops.conv2d(xt, conv, residual=rt, leaky=leaky, out=buf[:, 3:3 + co])
                assert bool((buf[:, :3] == 3).all()) and bool((buf[:, 3 + co:] == 3).all())
            else:
                y = ops.conv2d(xt, conv, residual=rt, leaky=leaky)
        ref = oracle.conv2d(x, conv.weight.detach().cpu().numpy(),
                            conv.bias.detach().cpu().numpy() if use_bias else None, stride=stride, residual=res,
                            leaky=leaky)
        assert tuple(y.shape) == ref.shape, (case, tuple(y.shape), ref.shape)
        assert_close(y, ref, tol=2e-5, what=f"case {case}: B{b} {ci}->{co} k{k} s{stride} {h}x{w} bias={use_bias} "
                                            f"res={use_res} leaky={leaky}")


@pytest.mark.parametrize("shape", [(2, 64, 64, 60, 80, True, 0.2), (1, 192, 64, 40, 48, True, 0.2), (3, 48, 48, 17, 29, False, None),
                                   (1, 384, 384, 15, 20, True, 0.2), (8, 64, 64, 120, 160, True, 0.2), (4, 32, 64, 37, 53, True, 0.0),
                                   (2, 128, 128, 64, 96, False, 0.2), (1, 64, 128, 8, 16, True, None), (5, 80, 64, 24, 40, True, 0.2),
                                   (3, 48, 32, 17, 29, True, 0.2), (8, 64, 64, 240, 320, True, 0.2), (2, 16, 64, 33, 47, True, 0.2)])
def test_winograd_pipeline_cases_and_work_order(shape, monkeypatch, sr_option):
    """The software-pipelined slab loop of sr_wino_kernel (next slab stored mid-slab, barrier after step 5, transform of
    its first channel group under the last MFMA steps; next REGION's first slab chained in when the slab count is even)
    over its structural cases: one slab (16 channels), odd and even slab counts (2, 3, 4, 5, 8, 12, 24), many regions
    per workgroup, fewer regions than CUs, split-K plans, ragged edges, Cout = 48 inside a 64-channel block and two
    64-channel blocks, with / without bias + residual, LeakyReLU / ReLU / identity.  The XCD-aware work order
    (SR_WINO_XCD) only permutes which workgroup computes which region: results are equal bit for bit, run to run too."""
    B, ci, co, h, w, extras, leaky = shape
    g = torch.Generator().manual_seed(ci + co + h)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1, bias=extras)
    x = torch.randn((B, ci, h, w), generator=g).to(DEV)
    res = torch.randn((B, co, h, w), generator=g).to(DEV) if extras else None
    conv = conv.to(DEV)
    sr_option("SR_CONV_WINO", 2)
    outs = []
    for mode in ("0", "1", "1"):
        sr_option("SR_WINO_XCD", int(mode))
        with torch.inference_mode():
            outs.append(ops.conv2d(x, conv, residual=res, leaky=leaky).clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    ref = torch.nn.functional.conv2d(x, conv.weight, conv.bias, padding=1)
    if res is not None:
        ref = ref + res
    if leaky is not None:
        ref = torch.nn.functional.leaky_relu(ref, leaky)
    assert rel_err(outs[1], ref) < 2e-5


@pytest.mark.parametrize("shape", [(8, 160, 960, 30, 40, "silu", False), (8, 960, 160, 30, 40, None, True),
                                   (2, 64, 128, 120, 160, None, False), (1, 192, 64, 48, 64, "relu", True),
                                   (3, 24, 1, 33, 47, None, False)])
def test_conv1x1_library_gemm_path(shape, monkeypatch):
    """1x1 convs over dense maps through hipBLASLt (sr_gemm1x1_nhwc_fwd: bias, BatchNorm fold, residual before the
    activation, SiLU / ReLU epilogues, channel-slice outputs) against F.conv2d and against the implicit-GEMM HIP kernel."""
    B, ci, co, h, w, act, with_res = shape
    g = torch.Generator().manual_seed(ci * 3 + co)
    conv = torch.nn.Conv2d(ci, co, 1).to(DEV)
    bn = torch.nn.BatchNorm2d(co).eval()
    synthetic.seeded_fill_(bn, seed=2)
    bn = bn.to(DEV)
    x = torch.randn((B, ci, h, w), generator=g).to(DEV)
    res = torch.randn((B, co, h, w), generator=g).to(DEV) if with_res else None
    kw = dict(act="silu") if act == "silu" else dict(leaky=0.0) if act == "relu" else {}
    ref = bn(torch.nn.functional.conv2d(x, conv.weight, conv.bias))
    if res is not None:
        ref = ref + res
    ref = torch.nn.functional.silu(ref) if act == "silu" else torch.relu(ref) if act == "relu" else ref
    outs = {}
    monkeypatch.setattr(ops, "USE_PW_1X1", False)   # (r04 default: the hand-written pointwise GEMM, tested below)
    for use in (True, False):
        monkeypatch.setattr(ops, "USE_GEMM_1X1", use)
        with torch.inference_mode():
            ops.PROFILE = []
            buf = ops.empty_nhwc(B, co + 8, h, w, DEV).fill_(3.0)      # write into a channel slice of a wider buffer
            ops.conv2d(x, conv, bn=bn, residual=res, out=buf[:, 4:4 + co], library_gemm=True, **kw)
            names = [r[0] for r in ops.PROFILE]
            ops.PROFILE = None
        assert ("hipBLASLt" in names[0]) == (use and B * h * w >= ops.GEMM_1X1_MIN_PIXELS), names
        assert bool((buf[:, :4] == 3).all()) and bool((buf[:, 4 + co:] == 3).all())
        outs[use] = buf[:, 4:4 + co].clone()
        assert rel_err(outs[use], ref.detach()) < 1e-5, use
    assert rel_err(outs[True], outs[False]) < 1e-5


PW_SHAPES = [  # (B, Cin, Cout, h, w, act, residual, gate)
    (8, 256, 1536, 15, 20, "silu", False, False),    # encoder stage 5 expand
    (8, 1536, 256, 15, 20, None, True, True),        # ... project: squeeze-excite gate + skip; K split across waves
    (8, 960, 160, 30, 40, None, True, True),         # stage 4 project (five 32-channel tiles)
    (2, 192, 64, 120, 160, None, False, False),      # BasicBlock skip
    (1, 112, 64, 48, 64, None, False, False),
    (3, 24, 48, 33, 47, "relu", True, False),        # ragged pixel tiles, Cout not a multiple of 32
    (2, 20, 1, 9, 11, None, False, False),           # Cin % 8 == 4, one output channel
    (1, 640, 384, 15, 20, 0.2, False, False),
    (2, 64, 100, 7, 5, "silu", True, True),          # 35 pixels per image: two tiles, the second ragged
]


@pytest.mark.parametrize("shape", PW_SHAPES)
@pytest.mark.parametrize("plan", [None, (0, 1), (0, 4), (1, 1), (1, 2), (2, 1), (3, 2), (3, 8)])
def test_pointwise_gemm_tiled_kernel(shape, plan, monkeypatch, sr_option):
    """sr_pw_conv_tiled_nhwc_fwd (csrc/sr_pw_tiled.hip): the LDS-tiled form of the same operator on batch-dense maps --
    every tile configuration (64x128, 128x160, 128x64, 64x64) and K split (partials added in index order), gate per image
    with pixel tiles that straddle images, ragged M / N / K tails, channel-slice operands -- against ATen in float64."""
    B, ci, co, h, w, act, with_res, with_gate = shape
    monkeypatch.setattr(ops, "PW_TILED", "1")
    if plan is not None:
        sr_option("SR_PT_CFG", plan[0])
        sr_option("SR_PT_KS", plan[1])
        if plan[1] > 1 and ((ci + 31) // 32 // plan[1] < 4 or co % 4):
            pytest.skip("K too short (or Cout not in float4 quads) for this split")
    ops._SHAPE_QUERIES.clear()
    g = torch.Generator().manual_seed(ci * 5 + co)
    conv = torch.nn.Conv2d(ci, co, 1).to(DEV)
    bn = synthetic.seeded_fill_(torch.nn.BatchNorm2d(co).eval(), seed=2).to(DEV)
    wide = torch.randn((B, ci + 8, h, w), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    x = wide[:, 4:4 + ci]
    res = torch.randn((B, co, h, w), generator=g).to(DEV) if with_res else None
    gate = torch.rand((B, ci), generator=g).to(DEV) if with_gate else None
    kw = dict(act="silu") if act == "silu" else dict(leaky=0.0) if act == "relu" else dict(leaky=act) if act else {}
    xr = x.double() * (gate.double()[:, :, None, None] if with_gate else 1.0)
    bnd = torch.nn.BatchNorm2d(co).eval().double().to(DEV)
    bnd.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    ref = bnd(torch.nn.functional.conv2d(xr, conv.weight.double(), conv.bias.double()))
    if res is not None:
        ref = ref + res.double()
    ref = torch.nn.functional.silu(ref) if act == "silu" else torch.relu(ref) if act == "relu" else \
        torch.nn.functional.leaky_relu(ref, act) if act else ref
    with torch.inference_mode():
        ops.PROFILE = []
        co_buf = (co + 3) // 4 * 4 + 8
        buf = ops.empty_nhwc(B, co_buf, h, w, DEV).fill_(3.0)
        ops.conv2d(x, conv, bn=bn, residual=res, out=buf[:, 4:4 + co], gate=gate, **kw)
        names = [r[0] for r in ops.PROFILE]
        ops.PROFILE = None
        again = ops.conv2d(x, conv, bn=bn, residual=res, gate=gate, **kw)
    ops._SHAPE_QUERIES.clear()
    assert names[0].startswith("sr_pw_tiled_kernel"), names
    if plan is not None:
        assert f"ks {plan[1]}," in names[0] and ("64x128", "128x160", "128x64", "64x64")[plan[0]] in names[0], names
    assert bool((buf[:, :4] == 3).all()) and bool((buf[:, 4 + co:] == 3).all())
    got = buf[:, 4:4 + co]
    assert rel_err(got, ref.float().detach()) < 1e-5
    assert torch.equal(got, again)      # fixed reduction order: bit-identical across calls and output layouts


@pytest.mark.parametrize("shape", PW_SHAPES)
@pytest.mark.parametrize("plan", [None, (1, 1), (2, 2), (2, 4), (4, 1)])
def test_pointwise_gemm_kernel(shape, plan, monkeypatch, sr_option):
    """sr_pw_conv_nhwc_fwd (csrc/sr_pw.hip): the 1x1 convolution as a hand-written fp32-MFMA GEMM -- bias, BatchNorm fold,
    residual before the activation, SiLU / ReLU / LeakyReLU, the squeeze-excite gate on the input, channel-slice inputs
    and outputs, every launch plan (channel tiles per wave, K split 1 / 2 / 4) -- against ATen in float64; deterministic."""
    B, ci, co, h, w, act, with_res, with_gate = shape
    monkeypatch.setattr(ops, "PW_TILED", "0")
    if plan is not None:
        sr_option("SR_PW_NT", plan[0])
        sr_option("SR_PW_KS", plan[1])
        if plan[1] > 1 and (ci + 7) // 8 // plan[1] < 8:
            pytest.skip("K too short for this split")
    g = torch.Generator().manual_seed(ci * 5 + co)
    conv = torch.nn.Conv2d(ci, co, 1).to(DEV)
    bn = synthetic.seeded_fill_(torch.nn.BatchNorm2d(co).eval(), seed=2).to(DEV)
    wide = torch.randn((B, ci + 8, h, w), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    x = wide[:, 4:4 + ci]                                     # a channel slice of a wider buffer (16-byte aligned rows)
    res = torch.randn((B, co, h, w), generator=g).to(DEV) if with_res else None
    gate = torch.rand((B, ci), generator=g).to(DEV) if with_gate else None
    kw = dict(act="silu") if act == "silu" else dict(leaky=0.0) if act == "relu" else dict(leaky=act) if act else {}
    xr = x.double() * (gate.double()[:, :, None, None] if with_gate else 1.0)
    bnd = torch.nn.BatchNorm2d(co).eval().double().to(DEV)
    bnd.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    ref = bnd(torch.nn.functional.conv2d(xr, conv.weight.double(), conv.bias.double()))
    if res is not None:
        ref = ref + res.double()
    ref = torch.nn.functional.silu(ref) if act == "silu" else torch.relu(ref) if act == "relu" else \
        torch.nn.functional.leaky_relu(ref, act) if act else ref
    with torch.inference_mode():
        ops.PROFILE = []
        co_buf = (co + 3) // 4 * 4 + 8
        buf = ops.empty_nhwc(B, co_buf, h, w, DEV).fill_(3.0)
        ops.conv2d(x, conv, bn=bn, residual=res, out=buf[:, 4:4 + co], gate=gate, **kw)
        names = [r[0] for r in ops.PROFILE]
        ops.PROFILE = None
        again = ops.conv2d(x, conv, bn=bn, residual=res, gate=gate, **kw)
    assert names[0].startswith("sr_pw_kernel"), names
    if plan is not None and not (plan[0] == 4 and plan[1] == 1 and False):
        assert names[0].startswith(f"sr_pw_kernel<{plan[0]}, {plan[1]}"), names
    assert bool((buf[:, :4] == 3).all()) and bool((buf[:, 4 + co:] == 3).all())
    got = buf[:, 4:4 + co]
    assert rel_err(got, ref.float().detach()) < 1e-5
    assert torch.equal(got, again)      # fixed reduction order: bit-identical across calls and output layouts


def test_pointwise_gemm_is_batch_independent_and_matches_the_conv_kernel(monkeypatch):
    """An image's result does not depend on what else is in the batch (pixel tiles never straddle images), and the GEMM agrees
    with the implicit-GEMM kernel it replaces."""
    g = torch.Generator().manual_seed(11)
    conv = torch.nn.Conv2d(128, 96, 1).to(DEV)
    x = torch.randn((5, 128, 17, 23), generator=g).to(DEV)
    with torch.inference_mode():
        outs = {}
        for tiled in ("0", "1"):
            monkeypatch.setattr(ops, "PW_TILED", tiled)
            all5 = ops.conv2d(x, conv, leaky=0.2)
            one = ops.conv2d(x[3:4].contiguous(memory_format=torch.channels_last), conv, leaky=0.2)
            assert torch.equal(all5[3:4], one), tiled    # (fp32 accumulation in k order per output: tiling does not matter)
            outs[tiled] = all5
        monkeypatch.setattr(ops, "USE_PW_1X1", False)
        old = ops.conv2d(x, conv, leaky=0.2)
    assert rel_err(outs["0"], old) < 1e-5 and rel_err(outs["1"], old) < 1e-5


@pytest.mark.parametrize("shape", [(1, 24, 240, 320, 24, "silu"), (2, 32, 240, 320, 24, None), (1, 16, 240, 320, 24, 0.2),
                                   (1, 24, 120, 168, 40, None)])
def test_winograd_many_regions_per_workgroup_with_a_channel_tail(shape):
    """Output channel counts that are not a multiple of 32 send EVERY region through the per-position ("border") epilogue of
    sr_wino_kernel; with more regions than workgroups each workgroup runs it many times in a row.  r04 regression: a 16-byte
    buffer store with an SGPR offset followed by a VALU write to its data registers stored the next store's OFFSET into
    channel 4 cg of a few pixels (the compiler does not know that hazard for register offsets) -- sparse, timing dependent,
    only visible at full resolution (EfficientNetV2-S stage 0 at 480x640)."""
    B, ci, H, W, co, act = shape
    g = torch.Generator().manual_seed(co)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
    x = torch.randn((B, ci, H, W), generator=g).to(DEV)
    res = torch.randn((B, co, H, W), generator=g).to(DEV)
    kw = dict(act="silu") if act == "silu" else dict(leaky=act) if act is not None else {}
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double()
    ref = torch.nn.functional.silu(ref) if act == "silu" else torch.nn.functional.leaky_relu(ref, act) if act is not None else ref
    with torch.inference_mode():
        for _ in range(3):
            out = ops.empty_nhwc(B, co, H, W, DEV).fill_(777.0)
            ops.conv2d(x, conv, residual=res, out=out, **kw)
            assert int((out == 777.0).sum()) == 0
            assert rel_err(out, ref.float()) < 1e-5
