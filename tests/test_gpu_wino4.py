"""GPU parity of the Winograd F(4x4, 3x3) kernel (csrc/sr_wino4.hip; conv3x3 + bias + residual + LeakyReLU of the reference's
BasicBlock, modules/layers.py:7-85) against an fp64 ATen convolution, the F(2x2) kernel and the direct implicit-GEMM kernel.
The tolerance is the fp32 bar of the north star relative to the output range, with the measured error printed."""
import ctypes as C

import pytest
import torch

from simplerecon_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(kind, x, conv, res, leaky, out=None):
    """kind: 'w4' F(4x4), 'w2' F(2x2), 'direct' -- each through its C-ABI entry point."""
    lib = _lib.lib()
    b, ci, h, w = x.shape
    co = conv.out_channels
    out = ops.empty_nhwc(b, co, h, w, x.device) if out is None else out
    isb, isp = ops._strides(x)
    osb, osp = ops._strides(out)
    rsb, rsp = ops._strides(res) if res is not None else (0, 0)
    slope = C.c_float(ops._act_code(leaky, None))
    with _lib.on_device(x.device):
        if kind in ("w4", "w4_pp", "w4_ws"):   # two 4-wave workgroups per CU / 8-wave ping-pong / wave-specialised 8-wave
            wp, bias = ops.packed_wino4_weight(conv)
            rc = lib.sr_conv3x3_wino4_variant_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(res), rsb,
                                                       rsp, _lib.ptr(out), osb, osp, b, h, w, ci, co, slope,
                                                       {"w4": 1, "w4_pp": 2, "w4_ws": 3}[kind], _lib.stream_ptr(x.device))
        elif kind == "w2":
            wp, bias = ops.packed_wino_weight(conv)
            rc = lib.sr_conv3x3_wino_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(res), rsb, rsp,
                                              _lib.ptr(out), osb, osp, b, h, w, ci, co, slope, _lib.stream_ptr(x.device))
        else:
            wp, bias = ops.packed_weight(conv)
            rc = lib.sr_conv2d_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(res), rsb, rsp,
                                        _lib.ptr(out), osb, osp, b, h, w, ci, co, 3, 1, slope, _lib.stream_ptr(x.device))
    _lib.check(rc, kind)
    return out


def _ref64(x, conv, res, leaky):
    y = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double() if conv.bias is not None else None,
                                   padding=1)
    if res is not None:
        y = y + res.double()
    return torch.nn.functional.leaky_relu(y, leaky) if leaky is not None else y


# (B, Cin, H, W, Cout, residual, bias): interior + border regions, ragged sizes, channel tails in Cin (not a multiple of 16)
# and Cout (not a multiple of 64 / 16), several items per workgroup, one region smaller than a tile
SHAPES = [(1, 16, 16, 16, 64, False, True), (1, 16, 16, 32, 64, True, True), (1, 16, 48, 16, 64, False, True),
          (1, 32, 33, 17, 64, True, True), (2, 16, 16, 16, 128, True, True), (5, 48, 16, 16, 64, True, True), (2, 64, 48, 64, 64, True, True), (1, 24, 35, 53, 64, True, False),
          (1, 64, 20, 18, 24, False, True), (2, 192, 32, 48, 64, True, True), (1, 128, 50, 70, 128, True, True),
          (1, 36, 9, 7, 12, True, True), (3, 64, 240, 320, 64, True, True), (1, 112, 17, 33, 72, False, False)]


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_wino4_matches_fp64_and_the_other_kernels(shape):
    b, ci, h, w, co, with_res, with_bias = shape
    torch.manual_seed(ci * 131 + co)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1, bias=with_bias).to(DEV)
    x = torch.randn(b, ci, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
    res = torch.randn(b, co, h, w, device=DEV).contiguous(memory_format=torch.channels_last) if with_res else None
    with torch.inference_mode():
        y4 = _run("w4", x, conv, res, 0.2)
        y4b = _run("w4_pp", x, conv, res, 0.2)
        y4c = _run("w4_ws", x, conv, res, 0.2)
        y2 = _run("w2", x, conv, res, 0.2)
        yd = _run("direct", x, conv, res, 0.2)
        ref = _ref64(x, conv, res, 0.2)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    e4 = (y4.double() - ref).abs().max().item() / scale
    e2 = (y2.double() - ref).abs().max().item() / scale
    ed = (yd.double() - ref).abs().max().item() / scale
    print(f"{shape}: rel-to-range error F(4x4) {e4:.2e}  F(2x2) {e2:.2e}  direct {ed:.2e}")
    assert e4 < 2e-5, f"F(4x4) error {e4} (F(2x2) {e2}, direct {ed})"
    assert torch.isfinite(y4).all()
    assert torch.equal(y4, y4b) and torch.equal(y4, y4c), "the kernel forms run the same operations in the same order"


@pytest.mark.parametrize("family", ["relu", "relu_dc3", "dc10", "dc100", "tails", "tails_relu", "gain8", "smooth"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 80, 64, True), (1, 192, 48, 64, 64, True)], ids=str)
def test_wino4_numerics_on_offset_and_heavy_tailed_inputs(family, shape):
    """F(4x4)'s error constant is 5-8x F(2x2)'s on zero-mean inputs and Winograd error grows with the DC offset / dynamic range of
    the input -- which is what the post-ReLU maps of the matching encoder's layer1 (reference modules/networks.py:176-182) and
    trained decoders feed it.  Families: tests/wino_numerics.py; measured table: profiles/r06_wino4_numerics.txt.  The bound is
    the same range-relative 2e-5 as on zero-mean inputs; the rms error must stay within 40x the F(2x2) kernel's (measured: <= 12x)
    so that a regression of the transform constants cannot hide under a wide output range."""
    import wino_numerics as WN
    b, ci, h, w, co, with_res = shape
    torch.manual_seed(ci + co)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
    with torch.no_grad():
        conv.weight.mul_(WN.weight_gain(family))
    x = WN.make_input(family, (b, ci, h, w), DEV).contiguous(memory_format=torch.channels_last)
    res = WN.make_input("relu" if "relu" in family else "randn", (b, co, h, w), DEV, seed=7).contiguous(
        memory_format=torch.channels_last) if with_res else None
    with torch.inference_mode():
        ref = _ref64(x, conv, res, 0.2)
        e = {}
        for kind in ("w4", "w4_ws", "w2", "direct"):
            e[kind] = WN.errors(_run(kind, x, conv, res, 0.2), ref)
        same = torch.equal(_run("w4", x, conv, res, 0.2), _run("w4_ws", x, conv, res, 0.2))
    torch.cuda.synchronize()
    print(f"{family} {shape}: range-rel F(4x4) {e['w4'][0]:.2e} F(2x2) {e['w2'][0]:.2e} direct {e['direct'][0]:.2e}; "
          f"rms-rel {e['w4'][1]:.2e} / {e['w2'][1]:.2e} / {e['direct'][1]:.2e}")
    assert same, "the kernel forms run the same operations in the same order"
    assert e["w4"][0] < 2e-5, f"F(4x4) range-relative error {e['w4'][0]} on `{family}` inputs (F(2x2) {e['w2'][0]}, direct {e['direct'][0]})"
    assert e["w4"][1] < 40 * max(e["w2"][1], 1e-8), f"F(4x4) rms error {e['w4'][1]} vs F(2x2) {e['w2'][1]}"


@pytest.mark.parametrize("shape", [(2, 64, 48, 64, 64, True), (1, 24, 35, 53, 64, True), (1, 128, 24, 40, 128, False)], ids=str)
def test_wino4_against_the_oracle_convolution(shape):
    """The same layer through the CPU oracle (oracle/sr_oracle.c, the restatement pinned against the reference's modules in
    tests/golden) rather than ATen: conv3x3 + bias + residual + LeakyReLU(0.2), fp64 accumulation, on post-ReLU inputs."""
    import numpy as np
    import wino_numerics as WN
    import oracle as O
    b, ci, h, w, co, with_res = shape
    torch.manual_seed(ci * 7 + co)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
    x = WN.make_input("relu", (b, ci, h, w), DEV).contiguous(memory_format=torch.channels_last)
    res = WN.make_input("randn", (b, co, h, w), DEV, seed=3).contiguous(memory_format=torch.channels_last) if with_res else None
    with torch.inference_mode():
        y4 = _run("w4_ws", x, conv, res, 0.2)
        ya = _ref64(x, conv, res, 0.2)
    ref = O.conv2d(x.cpu().numpy(), conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy(),
                   residual=None if res is None else res.cpu().numpy(), leaky=0.2, precision="f64")
    scale = float(np.abs(ref).max())
    err = float(np.abs(y4.cpu().double().numpy() - ref).max()) / scale
    aten = float(np.abs(ya.cpu().numpy() - ref).max()) / scale
    print(f"{shape}: F(4x4) vs oracle fp64 {err:.2e} of the range; ATen fp64 vs oracle fp64 {aten:.2e}")
    assert aten < 1e-12, "the two fp64 references disagree"
    assert err < 2e-5


def test_wino4_writes_into_a_concat_slice_and_reads_from_one():
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(32, 64, 3, padding=1).to(DEV)
    buf_in = ops.empty_nhwc(2, 80, 40, 48, DEV).normal_()
    x = buf_in[:, 16:48]
    buf_out = ops.empty_nhwc(2, 96, 40, 48, DEV).fill_(7.0)
    with torch.inference_mode():
        dense = _run("w4", (x * 1.0).contiguous(memory_format=torch.channels_last), conv, None, 0.2)
        _run("w4", x, conv, None, 0.2, out=buf_out[:, 16:80])
    torch.cuda.synchronize()
    assert torch.equal(buf_out[:, 16:80], dense)
    assert bool((buf_out[:, :16] == 7).all()) and bool((buf_out[:, 80:] == 7).all())


def test_wino4_is_deterministic_and_batch_independent():
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV)
    x = torch.randn(4, 64, 64, 80, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.inference_mode():
        a = _run("w4", x, conv, None, 0.2)
        b = _run("w4", x, conv, None, 0.2)
        one = _run("w4", x[2:3], conv, None, 0.2)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.equal(a[2:3], one)   # per-pixel arithmetic does not depend on the batch


def test_ws_form_under_concurrent_streams():
    """The wave-specialised form next to other kernels on side HIP streams (the decoder's branch parallelism at small batch,
    networks.py DepthDecoderPP): co-resident workgroups of OTHER kernels take issue slots on some SIMDs of a CU and skew the
    waves of a workgroup against each other; every output must still equal the quiet single-stream result of the 4-wave
    form bit for bit.  (r05: the first version handed the finished tile over through the V buffer a wave had just consumed
    while its sibling waves were still reading it -- run-to-run differences in tests/test_gpu_depth_model.py.)"""
    torch.manual_seed(9)
    shapes = [(2, 64, 120, 160, 64), (2, 128, 60, 80, 128), (1, 64, 240, 320, 64)]
    cases = []
    for (b, ci, h, w, co) in shapes:
        conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
        x = torch.randn(b, ci, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
        res = torch.randn(b, co, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
        with torch.inference_mode():
            want = _run("w4", x, conv, res, 0.2).clone()
            _run("w2", x, conv, res, 0.2)   # (packs the F(2x2) weights outside the concurrent part)
        cases.append((conv, x, res, want))
    noise = torch.randn(1 << 22, device=DEV)
    torch.cuda.synchronize()
    main, s1, s2 = (torch.cuda.Stream(device=DEV) for _ in range(3))
    outs = [[] for _ in cases]
    with torch.inference_mode():
        for rep in range(10):
            for i, (conv, x, res, _) in enumerate(cases):
                with torch.cuda.stream(s1):      # few-wave workgroups with little LDS: they fit next to the 8-wave workgroup
                    for _ in range(4):
                        noise = torch.sin(noise) * 1.0001
                with torch.cuda.stream(s2):
                    _run("w2", cases[(i + 1) % len(cases)][1], cases[(i + 1) % len(cases)][0], None, 0.2)
                with torch.cuda.stream(main):
                    outs[i].append(_run("w4_ws", x, conv, res, 0.2))
    torch.cuda.synchronize()
    for i, (_, _, _, want) in enumerate(cases):
        for rep, got in enumerate(outs[i]):
            assert torch.equal(got, want), f"shape {shapes[i]}, repetition {rep}: {int((got != want).sum())} elements differ"


@pytest.mark.parametrize("act", [None, 0.0, 0.2, "silu"])
def test_wino4_activation_codes(act):
    torch.manual_seed(8)
    conv = torch.nn.Conv2d(32, 64, 3, padding=1).to(DEV)
    x = torch.randn(2, 32, 40, 24, device=DEV).contiguous(memory_format=torch.channels_last)
    lib = _lib.lib()
    wp, bias = ops.packed_wino4_weight(conv)
    out = ops.empty_nhwc(2, 64, 40, 24, DEV)
    isb, isp = ops._strides(x)
    osb, osp = ops._strides(out)
    code = ops._act_code(None, "silu") if act == "silu" else ops._act_code(act, None)
    with torch.inference_mode():
        rc = lib.sr_conv3x3_wino4_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), None, 0, 0, _lib.ptr(out), osb,
                                           osp, 2, 40, 24, 32, 64, C.c_float(code), _lib.stream_ptr(x.device))
        _lib.check(rc, "sr_conv3x3_wino4_nhwc_fwd")
        y = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        ref = torch.nn.functional.silu(y) if act == "silu" else y if act is None else torch.nn.functional.leaky_relu(y, act)
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_wino4_non_finite_inputs_do_not_leak_across_channel_padding():
    """Channels past Cin inside the pixel stride (a concat buffer) must be masked, not multiplied by zero weights."""
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(24, 64, 3, padding=1).to(DEV)
    buf = ops.empty_nhwc(1, 40, 32, 32, DEV).normal_()
    buf[:, 24:] = float("nan")
    with torch.inference_mode():
        y = _run("w4", buf[:, :24], conv, None, 0.2)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()


def test_ops_conv2d_dispatches_wino4_by_rule_and_by_switch(monkeypatch):
    lib = _lib.lib()
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the rule's answers below are those of a 256-CU device (items per round of CUs)")
    assert lib.sr_conv_prefers_wino4(8, 240, 320, 64, 64, 1) == 3      # short slab chain: the wave-specialised form
    assert lib.sr_conv_prefers_wino4(8, 240, 320, 192, 64, 1) == 3     # long one: r05 the 4-wave form, since r06 the same
    assert lib.sr_conv_prefers_wino4(1, 240, 320, 64, 64, 1) == 0      # 300 items in two partial rounds: F(2x2)
    assert lib.sr_conv_prefers_wino4(8, 120, 160, 64, 64, 1) == 3      # 640 items = 2.5 rounds of 256 workgroups
    assert lib.sr_conv_prefers_wino4(8, 60, 80, 64, 64, 1) == 3        # one round on 160 CUs
    assert lib.sr_conv_prefers_wino4(8, 60, 80, 128, 128, 1) == 0      # 320 items: two partial rounds
    assert lib.sr_conv_prefers_wino4(8, 15, 20, 384, 384, 1) == 0
    assert lib.sr_conv_prefers_wino4(8, 240, 320, 64, 64, 0) == 0
    assert lib.sr_conv_prefers_wino4(1, 24, 24, 16, 16, 2) == 1
    torch.manual_seed(6)
    conv = torch.nn.Conv2d(32, 48, 3, padding=1).to(DEV)
    x = torch.randn(1, 32, 40, 40, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.inference_mode():
        monkeypatch.setattr(ops, "WINO4_MODE", 2)
        ops._SHAPE_QUERIES.clear()
        prof = []
        monkeypatch.setattr(ops, "PROFILE", prof)
        y = ops.conv2d(x, conv, leaky=0.2)
        assert prof[-1][0] in ("sr_wino4_kernel", "sr_wino4ws_kernel")
        monkeypatch.setattr(ops, "WINO4_MODE", 0)
        ops._SHAPE_QUERIES.clear()
        y0 = ops.conv2d(x, conv, leaky=0.2)
        assert not prof[-1][0].startswith("sr_wino4")
    ops._SHAPE_QUERIES.clear()
    assert (y - y0).abs().max().item() < 2e-5 * y0.abs().max().item()
