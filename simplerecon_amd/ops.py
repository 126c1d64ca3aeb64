"""Host-side plumbing for the HIP conv stack: channels-last tensor views, packed-weight cache,
and thin wrappers that hand raw pointers/strides to the C ABI (include/simplerecon_hip.h).

A "channels-last view" is a logically [B,C,H,W] torch tensor whose memory is [B,H,W,Ctot] with
this tensor occupying channels [c0, c0+C) of each pixel -- either a dense channels_last tensor or
a channel slice `buf[:, c0:c1]` of one.  Kernels take (pointer, batch stride, pixel stride)."""
import ctypes as C
import contextlib
import os
import weakref

import torch
from torch import nn

from . import _lib

# When set to a list, every conv launch appends (kernel symbol, algorithmic FLOPs, start event, end event):
# bench.py uses it to time the dominant kernel with HIP events on the launch stream.
PROFILE = None

_PACKED = weakref.WeakKeyDictionary()  # nn.Conv2d -> (weight version, data_ptr, packed tensor)

# Packed weights are produced by a kernel on whichever stream made the first call; a consumer on ANOTHER stream (the
# sub-batch streams of DepthModel.hot_path, the image-prior side stream) must not start before that kernel finished.
# (module, cache tag) -> (event recorded behind the pack kernel, its stream); the entry is dropped once the event completed.
_PACK_EVENTS = {}


def _packed_here(mod, tag, device):
    if device.type != "cuda" or not _lib.cuda_available():
        return
    st = torch.cuda.current_stream(device)
    if _lib.capturing():
        return  # weights are packed during warm-up, never inside a capture (graph.GraphedCallable warms up first)
    ev = torch.cuda.Event()
    ev.record(st)
    _PACK_EVENTS[(id(mod), tag)] = (ev, st.cuda_stream, weakref.ref(mod))


def _await_packed(mod, tag, device):
    if not _PACK_EVENTS:
        return
    hit = _PACK_EVENTS.get((id(mod), tag))
    if hit is None or hit[2]() is not mod:
        return
    if _lib.capturing():
        return
    ev, stream_ptr, _ = hit
    if ev.query():
        del _PACK_EVENTS[(id(mod), tag)]
        return
    st = torch.cuda.current_stream(device)
    if st.cuda_stream != stream_ptr:
        st.wait_event(ev)


def empty_nhwc(b, c, h, w, device):
    return torch.empty((b, c, h, w), dtype=torch.float32, device=device, memory_format=torch.channels_last)


def conv_out_hw(h, w, stride, ksize=3):
    pad = ksize // 2
    return (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1


def tf_same_pads(h, w, ksize, stride):
    """(top, left, bottom, right) zero padding of a TensorFlow-"SAME" convolution: output size ceil(i / stride),
    the odd pixel of padding goes below / right (what timm's Conv2dSame computes per call for its tf_* models)."""
    def one(i):
        total = max((-(-i // stride) - 1) * stride + ksize - i, 0)
        return total // 2, total - total // 2
    (pt, pb), (pl, pr) = one(h), one(w)
    return pt, pl, pb, pr


def _act_code(leaky, act):
    """The activation code the C ABI carries in `leaky_slope` (include/simplerecon_hip.h)."""
    if act is None:
        return -1.0 if leaky is None else float(leaky)
    if leaky is not None:
        raise ValueError("pass either `leaky` or `act`, not both")
    if act == "silu":
        return -2.0
    raise ValueError(f"unknown activation {act!r}")


def _is_nhwc_view(t):
    if t.dim() != 4:
        return False
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    if c == 1:
        sc = 1
    return sc == 1 and sw >= c and sh == w * sw and (b == 1 or sb >= h * sh)


def as_nhwc(t, name="tensor"):
    """Returns `t` if it already is a channels-last view, else a dense channels_last repack."""
    _lib.require_device_f32(name, t)
    if t.dim() != 4:
        raise ValueError(f"{name} must be [B,C,H,W], got {tuple(t.shape)}")
    if _is_nhwc_view(t):
        return t
    return t.contiguous(memory_format=torch.channels_last)


def _strides(t):
    """(batch stride, pixel stride) in elements of a channels-last view."""
    b, c, h, w = t.shape
    sb, _, _, sw = t.stride()
    if b == 1:
        sb = h * w * sw
    return sb, sw


def _state_key(conv, bn):
    # (version, address) of every tensor the packed weight is derived from; read through the modules' own dictionaries
    # (nn.Module.__getattr__ is a slow path and this runs once per launch)
    p = conv._parameters
    w, b = p["weight"], p.get("bias")
    if bn is None:
        if b is None:
            return (w._version, w.data_ptr())
        return (w._version, w.data_ptr(), b._version, b.data_ptr())
    ts = [w] if b is None else [w, b]
    bufs, bp = bn._buffers, bn._parameters
    ts += [bufs["running_mean"], bufs["running_var"]]
    if bn.affine:
        ts += [bp["weight"], bp["bias"]]
    return tuple((t._version, t.data_ptr()) if t is not None else None for t in ts)


def bn_affine(bn):
    """Eval-mode BatchNorm2d as the per-channel affine ATen's inference path applies:
    scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale."""
    if bn.running_mean is None or bn.running_var is None:
        raise _lib.HipLibraryError("BatchNorm without running statistics cannot run in inference mode")
    scale = torch.rsqrt(bn.running_var.detach() + bn.eps)
    if bn.affine:
        scale = scale * bn.weight.detach()
    shift = -bn.running_mean.detach() * scale
    if bn.affine:
        shift = shift + bn.bias.detach()
    return scale.contiguous(), shift.contiguous()


def _effective_weight(conv, bn):
    """(weight, bias) of `conv` with an eval-mode BatchNorm `bn` behind it folded in (one-off, at pack time)."""
    w = conv.weight.detach()
    b = conv.bias.detach() if conv.bias is not None else None
    if bn is not None:
        scale, shift = bn_affine(bn)
        w = w * scale.view(-1, 1, 1, 1)
        b = shift if b is None else b * scale + shift
    return w.contiguous(), b


def _check_conv(conv):
    k = conv.kernel_size[0]
    if conv.kernel_size[0] != conv.kernel_size[1] or k not in (1, 3) or conv.groups != 1 \
            or conv.dilation != (1, 1) or tuple(conv.padding) != (k // 2,) * 2 \
            or conv.padding_mode not in ("zeros", "replicate"):
        raise _lib.HipLibraryError(f"unsupported Conv2d configuration for the HIP path: {conv}")


def packed_weight(conv: nn.Conv2d, bn=None):
    """(packed weight, bias) for the direct kernel; cached until a parameter changes."""
    key = _state_key(conv, bn)
    hit = _PACKED.get(conv)
    if hit is not None and hit[0] == key:
        _await_packed(conv, "direct", hit[1].device)
        return hit[1], hit[2]
    _lib.require_device_f32("conv weight", conv.weight)
    _check_conv(conv)
    lib = _lib.lib()
    w, bias = _effective_weight(conv, bn)
    co, ci, k, _ = w.shape
    packed = torch.empty(lib.sr_conv_packed_weight_floats(co, ci, k), dtype=torch.float32, device=w.device)
    with _lib.on_device(w.device):
        rc = lib.sr_conv_pack_weights(_lib.ptr(w), co, ci, k, _lib.ptr(packed), _lib.stream_ptr(w.device))
    _lib.check(rc, "sr_conv_pack_weights")
    _packed_here(conv, "direct", w.device)
    _PACKED[conv] = (key, packed, bias)
    return packed, bias


_GEMM_W = weakref.WeakKeyDictionary()  # nn.Conv2d (1x1) -> (state key, [Cout, Cin] weight with BN folded, bias)
# 1x1 / stride-1 convolutions: "pw" (default) = the hand-written pointwise MFMA GEMM of csrc/sr_pw.hip (deterministic, r04);
# "lib" = hipBLASLt where the caller allows it (library_gemm=True; r02 / r03 behaviour: the algorithm is picked by timing,
# per process); "0" = the implicit-GEMM conv kernel for everything.
_MODE_1X1 = os.environ.get("SR_CONV1X1_GEMM", "pw")
USE_PW_1X1 = _MODE_1X1 not in ("0", "lib", "1")
# which of the two pointwise kernels: "auto" = the LDS-tiled one for batch-dense maps of at least PW_TILED_MIN_ROWS pixels
# (the HBM-bound full-resolution skips: 168 vs 182 us for 192 -> 64 @ 8x240x320), the direct one below that -- on the small-M
# encoder GEMMs the tiled kernel's one or two workgroups per CU cannot hide their load latency and it measured no faster
# (38 vs 36 us for 1536 -> 256 @ 8x15x20, profiles/r04_pw_plan_sweep.txt); "1" / "0" force one of them
PW_TILED = os.environ.get("SR_PW_TILED", "auto")
PW_TILED_MIN_ROWS = 100000
USE_GEMM_1X1 = _MODE_1X1 in ("lib", "1")
GEMM_1X1_MIN_PIXELS = 1024
SHORTCUT_GEMM = os.environ.get("SR_SHORTCUT_GEMM", "1") != "0"   # BasicBlock's 1x1 skip conv as a library GEMM


def gemm_weight(conv: nn.Conv2d, bn=None):
    """([Cout, Cin] weight, bias) of a 1x1 conv with the eval-mode BatchNorm folded in, for sr_gemm1x1_nhwc_fwd."""
    _lib.require_device_f32("conv weight", conv.weight)
    key = _state_key(conv, bn)
    hit = _GEMM_W.get(conv)
    if hit is not None and hit[0] == key:
        _await_packed(conv, "gemm", conv.weight.device)   # folded on another stream's first call?  wait for it
        return hit[1], hit[2]
    w, bias = _effective_weight(conv, bn)
    w2d = w.reshape(w.shape[0], w.shape[1]).contiguous()
    bias = bias.contiguous() if bias is not None else None
    _packed_here(conv, "gemm", conv.weight.device)
    _GEMM_W[conv] = (key, w2d, bias)
    return w2d, bias


_PACKED_LIN = weakref.WeakKeyDictionary()  # nn.Linear -> (weight version, data_ptr, packed tensor)


def linear(x, lin: nn.Linear, leaky=None):
    """act(x @ W^T + b) over the last dimension through the 1x1 instantiation of the HIP conv kernel
    ([M, Cin] row-major IS a 1 x M channels-last image).  Used by networks.MLP.forward."""
    _lib.require_device_f32("linear input", x)
    _lib.refuse_autograd(x, lin.weight)
    cin, cout = lin.in_features, lin.out_features
    if x.shape[-1] != cin:
        raise ValueError(f"linear expects {cin} input features, got {x.shape[-1]}")
    x2 = x.reshape(-1, cin).contiguous()
    m = x2.shape[0]
    out = torch.empty((m, cout), dtype=torch.float32, device=x.device)
    if m == 0:
        return out.view(*x.shape[:-1], cout)
    w = lin.weight
    hit = _PACKED_LIN.get(lin)
    lib = _lib.lib()
    if hit is not None and hit[0] == w._version and hit[1] == w.data_ptr():
        wp = hit[2]
        _await_packed(lin, "linear", w.device)
    else:
        wp = torch.empty(lib.sr_conv_packed_weight_floats(cout, cin, 1), dtype=torch.float32, device=w.device)
        with _lib.on_device(w.device):
            _lib.check(lib.sr_conv_pack_weights(_lib.ptr(w.detach().contiguous()), cout, cin, 1, _lib.ptr(wp),
                                                _lib.stream_ptr(w.device)), "sr_conv_pack_weights")
        _packed_here(lin, "linear", w.device)
        _PACKED_LIN[lin] = (w._version, w.data_ptr(), wp)
    bias = lin.bias.detach() if lin.bias is not None else None
    with _lib.on_device(x.device):
        rc = lib.sr_conv2d_nhwc_fwd(_lib.ptr(x2), m * cin, cin, _lib.ptr(wp), _lib.ptr(bias), None, 0, 0,
                                    _lib.ptr(out), m * cout, cout, 1, 1, m, cin, cout, 1, 1,
                                    C.c_float(-1.0 if leaky is None else float(leaky)), _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_conv2d_nhwc_fwd (linear)")
    return out.view(*x.shape[:-1], cout)


_PACKED_WINO = weakref.WeakKeyDictionary()  # nn.Conv2d -> (weight version, data_ptr, Winograd-packed tensor)


def wino_split_mode():
    """'' (off: fp32 MFMA, the product path), 'bf16' or 'f16': the fenced split-precision Winograd variant -- the library's
    option SR_WINO_SPLIT (one atomic load; set with _lib.set_option / experimental.split_precision)."""
    return _lib.split_mode_name("SR_WINO_SPLIT")


def packed_wino_weight(conv: nn.Conv2d, bn=None):
    """(Winograd-packed weight U = G g G^T, bias); cached until a parameter changes."""
    key = (_state_key(conv, bn), wino_split_mode())   # (the fenced split-precision variant packs 16-bit pieces)
    hit = _PACKED_WINO.get(conv)
    if hit is not None and hit[0] == key:
        _await_packed(conv, "wino", hit[1].device)
        return hit[1], hit[2]
    _check_conv(conv)
    lib = _lib.lib()
    w, bias = _effective_weight(conv, bn)
    co, ci = w.shape[:2]
    packed = torch.empty(lib.sr_wino_packed_weight_floats(co, ci), dtype=torch.float32, device=w.device)
    with _lib.on_device(w.device):
        rc = lib.sr_wino_pack_weights(_lib.ptr(w), co, ci, _lib.ptr(packed), _lib.stream_ptr(w.device))
    _lib.check(rc, "sr_wino_pack_weights")
    _packed_here(conv, "wino", w.device)
    _PACKED_WINO[conv] = (key, packed, bias)
    return packed, bias


_PACKED_WINO4 = weakref.WeakKeyDictionary()  # nn.Conv2d -> (state key, F(4x4, 3x3)-packed tensor, bias)
# Which 3x3 / stride-1 layers take the F(4x4, 3x3) kernel (csrc/sr_wino4.hip): "1" (default) = the library's rule
# (sr_conv_prefers_wino4: the full-resolution layers at batch 8), "0" = none, "2" = every layer it applies to (tests).
# Read once, handed to the library as an argument.
WINO4_MODE = int(os.environ.get("SR_CONV_WINO4", "1"))
# kernel form: 0 = the library's default, 1 = two independent 4-wave workgroups per CU, 2 = the 8-wave ping-pong workgroup
# (bit-identical results; A/B measurements)
WINO4_VARIANT = int(os.environ.get("SR_WINO4_VARIANT", "0"))


def packed_wino4_weight(conv: nn.Conv2d, bn=None):
    """(F(4x4, 3x3)-packed weight U = G g G^T, bias); cached until a parameter changes."""
    key = _state_key(conv, bn)
    hit = _PACKED_WINO4.get(conv)
    if hit is not None and hit[0] == key:
        _await_packed(conv, "wino4", hit[1].device)
        return hit[1], hit[2]
    _check_conv(conv)
    lib = _lib.lib()
    w, bias = _effective_weight(conv, bn)
    co, ci = w.shape[:2]
    packed = torch.empty(lib.sr_wino4_packed_weight_floats(co, ci), dtype=torch.float32, device=w.device)
    with _lib.on_device(w.device):
        rc = lib.sr_wino4_pack_weights(_lib.ptr(w), co, ci, _lib.ptr(packed), _lib.stream_ptr(w.device))
    _lib.check(rc, "sr_wino4_pack_weights")
    _packed_here(conv, "wino4", w.device)
    _PACKED_WINO4[conv] = (key, packed, bias)
    return packed, bias


# Pure functions of the layer shape inside the C library (launch-plan choices): asked once per shape, not once per launch.
_SHAPE_QUERIES = {}


def _drop_shape_queries(*_):
    _SHAPE_QUERIES.clear()


# launch plans depend on the forcing switches (SR_PT_*, SR_PW_*, SR_CONV_WINO ...): any option change drops the cached answers
_lib.OPTION_LISTENERS.append(_drop_shape_queries)


def _shape_query(lib, name, *shape, dev=None):
    # `dev`: device index of the tensors (the answers may depend on the device's CU count: sr_conv_prefers_wino4, split-K plans)
    key = (name, shape, dev)
    v = _SHAPE_QUERIES.get(key)
    if v is None:
        v = _SHAPE_QUERIES[key] = getattr(lib, name)(*shape)
    return v


def conv2d(x, conv: nn.Conv2d, residual=None, leaky=None, out=None, bn=None, act=None, tf_same=False, library_gemm=False,
           gate=None):
    """act(bn(conv(x) + bias) [+ residual]) with nn.Conv2d semantics (zero or replicate padding); `bn` is an
    eval-mode BatchNorm2d folded into weight and bias.  `leaky` = LeakyReLU slope, or act="silu".  tf_same=True
    replaces the module's symmetric padding by TensorFlow-"SAME" padding (tf_same_pads).  library_gemm=True lets a 1x1
    conv over a dense map run as a hipBLASLt GEMM (faster on the MBConv shapes; its algorithm choice depends on the number
    of pixels; only with SR_CONV1X1_GEMM=lib since r04).  Batch-size invariance: every kernel on the default path is run-to-run
    deterministic, but the RESULT OF A FRAME MAY DEPEND ON THE BATCH IT IS IN (and on the CU count of the device) in its low-order
    bits: the launch plans are chosen from the total number of work items, B included -- a 3x3 / stride-1 layer runs F(4x4)
    Winograd at one batch size and F(2x2) at another (sr_conv_prefers_wino4: 64 -> 64 at 240x320 is F(2x2) at B = 1, F(4x4) at
    B = 8; error constants 3e-7 vs 1.3e-6 of the output range), and the Winograd / direct / pointwise plans split K on small
    maps.  Within ONE plan the per-pixel arithmetic does not depend on B.  tests/test_gpu_determinism.py pins the run-to-run
    statement and the 1x1 cases, tests/test_gpu_e2e_full_size.py::test_batch_1_and_batch_8_agree_frame_by_frame the model-level
    agreement (to the element-wise tolerance each holds against the oracle, not bitwise).  `gate` ([B, Cin], 1x1 / stride-1 convs only): the input is scaled
    per image and input channel while it is loaded (the squeeze-excite gate of an MBConv block).  Returns a channels-last view."""
    _lib.refuse_autograd(x, conv.weight)
    x = as_nhwc(x, "conv input")
    b, ci, h, w = x.shape
    if ci != conv.in_channels:
        raise ValueError(f"conv expects {conv.in_channels} input channels, got {ci}")
    k, s = conv.kernel_size[0], conv.stride[0]
    if conv.stride[0] != conv.stride[1] or s not in (1, 2):
        raise _lib.HipLibraryError(f"unsupported stride {conv.stride}")
    pads = tf_same_pads(h, w, k, s) if tf_same else (k // 2,) * 4
    ho, wo = (h + pads[0] + pads[2] - k) // s + 1, (w + pads[1] + pads[3] - k) // s + 1
    co = conv.out_channels
    if out is None:
        out = empty_nhwc(b, co, ho, wo, x.device)
    else:
        if tuple(out.shape) != (b, co, ho, wo) or not _is_nhwc_view(out):
            raise ValueError(f"`out` must be a channels-last view of shape {(b, co, ho, wo)}")
        _lib.require_device_f32("out", out)
    lib = _lib.lib()
    replicate = conv.padding_mode == "replicate"
    padded = pads != (k // 2,) * 4
    if padded and replicate:
        raise _lib.HipLibraryError("explicit padding is implemented for zero padding only")
    use_wino = (not replicate) and (not padded) and bool(_shape_query(lib, "sr_conv_prefers_wino", b, h, w, ci, co, k, s, dev=x.device.index))
    if residual is not None:
        residual = as_nhwc(residual, "residual")
        if tuple(residual.shape) != (b, co, ho, wo):
            raise ValueError(f"residual shape {tuple(residual.shape)} != output shape {(b, co, ho, wo)}")
    if b == 0:
        return out
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    rsb, rsp = _strides(residual) if residual is not None else (0, 0)
    prof = PROFILE
    if use_wino and wino_split_mode():
        # fenced split-precision Winograd (SR_WINO_SPLIT, csrc/sr_wino_split.hip): the vector instantiation only -- layers with
        # unaligned tensors or channel counts that are no multiple of 4 take the direct fp32 kernel instead
        al = lambda t, sb_, sp_: t is None or (t.data_ptr() % 16 == 0 and sp_ % 4 == 0 and sb_ % 4 == 0)
        if not (ci % 4 == 0 and co % 4 == 0 and al(x, isb, isp) and al(out, osb, osp) and al(residual, rsb, rsp)):
            use_wino = False
    if gate is not None:
        _lib.require_device_f32("gate", gate)
        if k != 1 or s != 1 or tuple(gate.shape) != (b, ci) or not gate.is_contiguous():
            raise ValueError(f"`gate` needs a 1x1 / stride-1 conv and a contiguous [{b}, {ci}] tensor")
    if k == 1 and s == 1 and not padded and not replicate and (USE_PW_1X1 or gate is not None) and ci % 4 == 0 \
            and x.data_ptr() % 16 == 0 and isp % 4 == 0 and isb % 4 == 0:
        wp, bias = packed_weight(conv, bn)
        m = b * h * w
        dense = (b == 1 or (isb == h * w * isp and osb == h * w * osp and (residual is None or rsb == h * w * rsp)))
        tiled = dense and (PW_TILED == "1" or (PW_TILED == "auto" and m >= PW_TILED_MIN_ROWS))
        gflag = 'true' if gate is not None else 'false'
        with _lib.on_device(x.device):
            if tiled:
                nbytes = _shape_query(lib, "sr_pw_conv_tiled_workspace_bytes", m, ci, co, dev=x.device.index)
                ws = _workspace(x.device, "pw_splitk", nbytes) if nbytes else None
            if prof is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if tiled:
                rc = lib.sr_pw_conv_tiled_nhwc_fwd(_lib.ptr(x), isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(gate),
                                                   _lib.ptr(residual), rsp, _lib.ptr(out), osp, m, h * w, ci, co,
                                                   C.c_float(_act_code(leaky, act)), _lib.ptr(ws), nbytes,
                                                   _lib.stream_ptr(x.device))
            else:
                rc = lib.sr_pw_conv_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(gate),
                                             _lib.ptr(residual), rsb, rsp, _lib.ptr(out), osb, osp, b, h * w, ci, co,
                                             C.c_float(_act_code(leaky, act)), _lib.stream_ptr(x.device))
            if prof is not None:
                ev1.record()
                a_, b_ = C.c_int(0), C.c_int(0)
                if tiled:
                    # the plan the LAUNCHER took: it splits K only with a usable workspace AND 16-byte aligned out / residual /
                    # bias rows (sr_pw_conv_tiled_nhwc_fwd's `can_split`), so ask with the same answer
                    al16 = lambda t, sp_: t is None or (t.data_ptr() % 16 == 0 and sp_ % 4 == 0)
                    can_split = ws is not None and ws.data_ptr() % 16 == 0 and co % 4 == 0 and al16(out, osp) and \
                        al16(residual, rsp) and (bias is None or bias.data_ptr() % 16 == 0)
                    lib.sr_pw_conv_tiled_plan(m, ci, co, int(can_split), C.byref(a_), C.byref(b_))
                    if b_.value > 1 and nbytes < b_.value * m * co * 4:
                        lib.sr_pw_conv_tiled_plan(m, ci, co, 0, C.byref(a_), C.byref(b_))
                    tile_m, tile_n = ((64, 128), (128, 160), (128, 64), (64, 64))[a_.value]
                    executed = 2.0 * (-(-m // tile_m) * tile_m) * (-(-co // tile_n) * tile_n) * (-(-ci // 32) * 32)
                    name = f"sr_pw_tiled_kernel<{tile_m}x{tile_n}, ks {b_.value}, {gflag}>"
                else:
                    lib.sr_pw_conv_plan(b, h * w, ci, co, C.byref(a_), C.byref(b_))
                    mt = b * ((h * w + 31) // 32)
                    executed = 2.0 * mt * 32 * ((co + 32 * a_.value - 1) // (32 * a_.value) * 32 * a_.value) * ((ci + 7) // 8 * 8)
                    name = f"sr_pw_kernel<{a_.value}, {b_.value}, {gflag}>"
                prof.append((name, 2.0 * b * h * w * co * ci, ev0, ev1, (b, ci, h, w, co, k, s, ho, wo, residual is not None),
                             executed))
        if rc != 2 or gate is not None:   # SR_ERR_UNSUPPORTED (a per-image byte range past 2^31): the implicit-GEMM kernel below,
            _lib.check(rc, "sr_pw_conv_tiled_nhwc_fwd" if tiled else "sr_pw_conv_nhwc_fwd")   # whose addressing is 64-bit
            return out
        if prof is not None:
            prof.pop()
    if gate is not None:
        raise _lib.HipLibraryError("a gated 1x1 convolution needs Cin % 4 == 0 and 16-byte aligned input rows")
    if k == 1 and s == 1 and (w % 32 != 0 or h % 4 != 0) and (isp, isb) == (ci, h * w * ci) \
            and (osp, osb) == (co, h * w * co) and (residual is None or (rsp, rsb) == (co, h * w * co)):
        # a 1x1 conv over dense channels-last maps is a [B*H*W, Cin] x [Cin, Cout] product: hand the kernel the
        # pixels as one 32- (or 8-) wide strip so that no output tile is cut at image borders (15x20 maps would
        # leave 37 % of every 4x32 tile empty).  Same per-pixel arithmetic, same result.
        m = b * h * w
        fw = 32 if m % 32 == 0 else 8 if m % 8 == 0 else 0
        if fw:
            b, h, w, ho, wo = 1, m // fw, fw, m // fw, fw
            isb, osb, rsb = m * ci, m * co, (m * co if residual is not None else 0)
    # 1x1 convs over dense maps are plain GEMMs: hipBLASLt (north star: "rocBLAS/MFMA only where it is a dense im2col
    # GEMM") is 1.25-1.75x faster than the implicit-GEMM kernel on the MBConv shapes (scripts/gemm_probe.py)
    if library_gemm and k == 1 and s == 1 and not padded and not replicate and USE_GEMM_1X1 and \
            b * h * w >= GEMM_1X1_MIN_PIXELS:
        gact = 0 if (leaky is None and act is None) else 1 if act == "silu" else 2 if (act is None and leaky == 0.0) else -1
        dense = (isb == h * w * isp or b == 1) and (osb == ho * wo * osp or b == 1) and \
            (residual is None or rsb == ho * wo * rsp or b == 1)
        if gact >= 0 and dense:
            w2d, gbias = gemm_weight(conv, bn)
            ws = _workspace(x.device, "gemm1x1", lib.sr_gemm1x1_workspace_bytes())
            with _lib.on_device(x.device):
                if prof is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                rc = lib.sr_gemm1x1_nhwc_fwd(_lib.ptr(x), isp, _lib.ptr(w2d), _lib.ptr(gbias), _lib.ptr(residual), rsp,
                                             _lib.ptr(out), osp, b * h * w, ci, co, gact, _lib.ptr(ws), ws.numel() * 4,
                                             _lib.stream_ptr(x.device))
                if prof is not None and rc == 0:
                    ev1.record()
                    prof.append(("hipBLASLt fp32 GEMM (1x1 conv)", 2.0 * b * ho * wo * co * ci, ev0, ev1,
                                 (b, ci, h, w, co, k, s, ho, wo, residual is not None), None))
            if rc == 0:
                return out
            if rc != 2:   # SR_ERR_UNSUPPORTED: no library algorithm for this shape -> the HIP kernel below
                _lib.check(rc, "sr_gemm1x1_nhwc_fwd")
    w4_form = _shape_query(lib, "sr_conv_prefers_wino4", b, h, w, ci, co, WINO4_MODE, dev=x.device.index) \
        if (use_wino and WINO4_MODE and not wino_split_mode()) else 0   # 0: F(2x2); 1 / 3: the kernel form the rule picks
    if w4_form:
        al = lambda t, sb_, sp_: t is None or (t.data_ptr() % 16 == 0 and sp_ % 4 == 0 and sb_ % 4 == 0)
        wp4, bias4 = packed_wino4_weight(conv, bn)
        if al(x, isb, isp) and al(out, osb, osp) and al(residual, rsb, rsp) and (bias4 is None or bias4.data_ptr() % 16 == 0):
            with _lib.on_device(x.device):
                if prof is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                rc = lib.sr_conv3x3_wino4_variant_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp4), _lib.ptr(bias4),
                                                           _lib.ptr(residual), rsb, rsp, _lib.ptr(out), osb, osp, b, h, w, ci, co,
                                                           C.c_float(_act_code(leaky, act)), WINO4_VARIANT or w4_form,
                                                           _lib.stream_ptr(x.device))
                if prof is not None and rc == 0:
                    ev1.record()
                    regions = ((h + 15) // 16) * ((w + 15) // 16)   # multiplies issued: 36 per 4x4 tile and (ci, co) pair, padded
                    executed = 2.0 * b * regions * 16 * 36 * ((ci + 15) // 16 * 16) * ((co + 63) // 64 * 64)
                    prof.append(("sr_wino4ws_kernel" if (WINO4_VARIANT or w4_form) == 3 else "sr_wino4_kernel",
                                 2.0 * b * ho * wo * co * ci * 9, ev0, ev1,
                                 (b, ci, h, w, co, k, s, ho, wo, residual is not None), executed))
            if rc == 0:
                return out
            if rc != 2:   # SR_ERR_UNSUPPORTED (per-image byte range): the F(2x2) kernel below
                _lib.check(rc, "sr_conv3x3_wino4_nhwc_fwd")
    # (pack and launch must see ONE split mode: ADVICE r05.  A plain acquire / release -- a context manager object per call costs
    # a microsecond on a launch path that is host-bound at batch 1)
    if use_wino:
        _lib.SPLIT_GUARD.acquire()
    try:
        wp, bias = packed_wino_weight(conv, bn) if use_wino else packed_weight(conv, bn)
        slope = C.c_float(_act_code(leaky, act))
        with _lib.on_device(x.device):
            if prof is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if padded:
                rc = lib.sr_conv2d_padded_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(residual),
                                                   rsb, rsp, _lib.ptr(out), osb, osp, b, h, w, ci, co, k, s, *pads, slope,
                                                   _lib.stream_ptr(x.device))
            elif use_wino:
                nbytes = _shape_query(lib, "sr_wino_splitk_workspace_bytes", b, h, w, ci, co, dev=x.device.index)   # 0 unless the plan splits K
                ws = _workspace(x.device, "wino_splitk", nbytes) if nbytes else None
                rc = lib.sr_conv3x3_wino_splitk_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias),
                                                         _lib.ptr(residual), rsb, rsp, _lib.ptr(out), osb, osp, b, h, w, ci,
                                                         co, slope, _lib.ptr(ws), nbytes, _lib.stream_ptr(x.device))
            elif replicate:
                rc = lib.sr_conv2d_replicate_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias),
                                                      _lib.ptr(residual), rsb, rsp, _lib.ptr(out), osb, osp, b, h, w, ci,
                                                      co, k, s, slope, _lib.stream_ptr(x.device))
            else:
                nbytes = _shape_query(lib, "sr_conv_splitk_workspace_bytes", b, h, w, ci, co, k, s, dev=x.device.index)   # 0 unless it may split K
                ws = _workspace(x.device, "conv_splitk", nbytes) if nbytes else None
                rc = lib.sr_conv2d_splitk_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(residual),
                                                   rsb, rsp, _lib.ptr(out), osb, osp, b, h, w, ci, co, k, s, slope,
                                                   _lib.ptr(ws), nbytes, _lib.stream_ptr(x.device))
            if prof is not None:
                ev1.record()
                v4 = int(x.data_ptr() % 16 == 0 and isp % 4 == 0 and isb % 4 == 0)
                if use_wino:
                    vin = bool(v4 and ci % 4 == 0)
                    al = lambda t, sb_, sp_: t is None or (t.data_ptr() % 16 == 0 and sp_ % 4 == 0 and sb_ % 4 == 0)
                    vout = vin and co % 4 == 0 and al(out, osb, osp) and al(residual, rsb, rsp) and \
                        (bias is None or bias.data_ptr() % 16 == 0)
                    name = lib.sr_wino_kernel_name(b, h, w, ci, co, int(vin), int(vout)).decode()
                    if wino_split_mode():   # (the fenced variant: sr_wino_split_kernel<NT, FMT>)
                        name = name.replace("sr_wino_kernel<", "sr_wino_split_kernel<").replace(
                            ", true, true>", f", {1 if wino_split_mode() == 'bf16' else 2}>")
                else:
                    name = lib.sr_conv_kernel_name(b, h, w, ci, co, k, s, v4).decode()
                executed = None
                if use_wino:  # multiplies actually issued: 16 per 2x2 tile and (ci, co) pair, on padded regions / channels
                    regions = ((h + 7) // 8) * ((w + 15) // 16)
                    executed = 2.0 * b * regions * 32 * 16 * ((ci + 15) // 16 * 16) * ((co + 31) // 32 * 32)
                prof.append((name, 2.0 * b * ho * wo * co * ci * k * k, ev0, ev1,
                             (b, ci, h, w, co, k, s, ho, wo, residual is not None), executed))
        if use_wino and rc == 2 and not wino_split_mode():
            # SR_ERR_UNSUPPORTED from the Winograd entry point (a per-image byte range past 2^31: its buffer descriptors carry 32-bit
            # offsets): the implicit-GEMM kernel serves the shape with 64-bit addressing, as it did before r04
            if prof is not None:
                prof.pop()
            wp, bias = packed_weight(conv, bn)
            with _lib.on_device(x.device):
                rc = lib.sr_conv2d_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(residual), rsb, rsp,
                                            _lib.ptr(out), osb, osp, b, h, w, ci, co, k, s, slope, _lib.stream_ptr(x.device))
            _lib.check(rc, "sr_conv2d_nhwc_fwd")
            return out
        _lib.check(rc, "sr_conv3x3_wino_nhwc_fwd" if use_wino else "sr_conv2d_nhwc_fwd")
        return out
    finally:
        if use_wino:
            _lib.SPLIT_GUARD.release()


def basic_block(block, x, out=None):
    """BasicBlock.forward of the reference (modules/layers.py:68-85) as 2 (or 3) fused launches."""
    if not isinstance(block.bn1, nn.Identity) or not isinstance(block.bn2, nn.Identity):
        raise _lib.HipLibraryError("the HIP BasicBlock implements norm_layer=nn.Identity only (what SimpleRecon uses)")
    slope = block.relu.negative_slope
    x = as_nhwc(x, "BasicBlock input")
    t = conv2d(x, block.conv1, leaky=slope)
    identity = x if block.downsample is None else conv2d(x, block.downsample[0], library_gemm=SHORTCUT_GEMM)
    return conv2d(t, block.conv2, residual=identity, leaky=slope, out=out)


def upsample2x(x, out=None):
    """Bilinear x2, align_corners=False (reference utils/generic_utils.py:96-105)."""
    x = as_nhwc(x, "upsample input")
    _lib.refuse_autograd(x)
    b, c, h, w = x.shape
    if out is None:
        out = empty_nhwc(b, c, 2 * h, 2 * w, x.device)
    elif tuple(out.shape) != (b, c, 2 * h, 2 * w) or not _is_nhwc_view(out):
        raise ValueError(f"`out` must be a channels-last view of shape {(b, c, 2 * h, 2 * w)}")
    if b == 0:
        return out
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    with _lib.on_device(x.device):
        rc = _lib.lib().sr_upsample2x_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(out), osb, osp, b, h, w, c,
                                               _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_upsample2x_nhwc_fwd")
    return out


def exp(x):
    """Elementwise exp on a dense fp32 device tensor (any memory format; the result has the same strides)."""
    _lib.require_device_f32("exp input", x)
    if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
        x = x.contiguous()
    out = torch.empty_like(x)
    with _lib.on_device(x.device):
        rc = _lib.lib().sr_exp_fwd(_lib.ptr(x), _lib.ptr(out), x.numel(), _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_exp_fwd")
    return out


def copy_into(dst_view, src):
    """Copies a [B,C,H,W] tensor (any layout) into a channels-last slice (device-side strided copy)."""
    _lib.require_device_f32("copy source", src)
    dst_view.copy_(src)
    return dst_view


# ---------------------------------------------------------------- matching encoder ops --

_PACKED_STEM = weakref.WeakKeyDictionary()  # nn.Conv2d -> (state key, packed weight, scale, shift)
_WORKSPACES = {}                            # (device, tag) -> scratch tensor (grown on demand)


def _workspace(device, tag, nbytes):
    # one scratch buffer per (device, purpose, stream): launches on a stream are ordered, different streams must not share
    if _lib.capturing():
        # a HIP graph bakes the pointer in: give it memory from its own pool, never a cached buffer that a later eager
        # call may grow and drop
        return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    key = (device, tag, _lib.stream_id(device))
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WORKSPACES[key] = ws
    return ws


def stem7x7(image, conv: nn.Conv2d, bn=None, leaky=0.0, out=None):
    """act(bn(conv7x7_s2_p3(image))): encoder.conv1 + bn1 + relu of the ResNet stem (reference networks.py:176-179).
    image [B,3,H,W], any strides; returns channels-last [B,64,H/2,W/2] (or writes it into `out`, e.g. a batch slice
    of a larger buffer)."""
    _lib.require_device_f32("image", image)
    _lib.refuse_autograd(image, conv.weight)
    if conv.kernel_size != (7, 7) or conv.stride != (2, 2) or tuple(conv.padding) != (3, 3) or conv.groups != 1 \
            or conv.dilation != (1, 1) or conv.in_channels != 3 or conv.out_channels != 64 \
            or conv.padding_mode != "zeros":
        raise _lib.HipLibraryError(f"the HIP stem implements Conv2d(3, 64, 7, stride=2, padding=3) only, got {conv}")
    if image.dim() != 4 or image.shape[1] != 3:
        raise ValueError(f"stem expects [B,3,H,W], got {tuple(image.shape)}")
    _lib.require_device_f32("stem weight", conv.weight)
    lib = _lib.lib()
    key = _state_key(conv, bn)
    hit = _PACKED_STEM.get(conv)
    if hit is None or hit[0] != key:
        wp = torch.empty(lib.sr_stem_packed_weight_floats(64), dtype=torch.float32, device=conv.weight.device)
        with _lib.on_device(conv.weight.device):
            _lib.check(lib.sr_stem_pack_weights(_lib.ptr(conv.weight.detach().contiguous()), 64, _lib.ptr(wp),
                                                _lib.stream_ptr(conv.weight.device)), "sr_stem_pack_weights")
        scale = shift = None
        if bn is not None:
            scale, shift = bn_affine(bn)
        if conv.bias is not None:
            cb = conv.bias.detach()
            shift = (cb if scale is None else cb * scale) + (0 if shift is None else shift)
            shift = shift.contiguous()
        hit = (key, wp, scale, shift)
        _packed_here(conv, "stem", conv.weight.device)
        _PACKED_STEM[conv] = hit
    else:
        _await_packed(conv, "stem", conv.weight.device)
    _, wp, scale, shift = hit
    b, _, h, w = image.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if out is None:
        out = empty_nhwc(b, 64, ho, wo, image.device)
    elif tuple(out.shape) != (b, 64, ho, wo) or not _is_nhwc_view(out):
        raise ValueError(f"`out` must be a channels-last view of shape {(b, 64, ho, wo)}")
    if b == 0:
        return out
    sb, sc, sy, sx = image.stride()
    osb, osp = _strides(out)
    prof = PROFILE
    with _lib.on_device(image.device):
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = lib.sr_stem7x7_fwd(_lib.ptr(image), sb, sc, sy, sx, _lib.ptr(wp), _lib.ptr(scale), _lib.ptr(shift),
                                C.c_float(-1.0 if leaky is None else float(leaky)), _lib.ptr(out), osb, osp, b, h, w,
                                64, _lib.stream_ptr(image.device))
        if prof is not None:
            ev1.record()
            prof.append(("sr_stem_kernel", 2.0 * b * ho * wo * 64 * 147, ev0, ev1, (b, 3, h, w, 64, 7, 2),
                         2.0 * b * ((ho + 15) // 16) * ((wo + 15) // 16) * 256 * 64 * 148))
    _lib.check(rc, "sr_stem7x7_fwd")
    return out


def maxblurpool(x, out=None):
    """nn.MaxPool2d(2, stride=1) + antialiased_cnns.BlurPool(filt_size=4, stride=2), fused; channels-last (`out`: a
    channels-last view to write into, e.g. a channel slice of a larger buffer)."""
    x = as_nhwc(x, "maxblurpool input")
    b, c, h, w = x.shape
    if h < 4 or w < 4:
        raise ValueError(f"maxblurpool needs H, W >= 4, got {(h, w)}")
    ho, wo = (h - 2) // 2 + 1, (w - 2) // 2 + 1
    if out is None:
        out = empty_nhwc(b, c, ho, wo, x.device)
    elif tuple(out.shape) != (b, c, ho, wo) or not _is_nhwc_view(out):
        raise ValueError(f"`out` must be a channels-last view of shape {(b, c, ho, wo)}")
    if b == 0:
        return out
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    with _lib.on_device(x.device):
        rc = _lib.lib().sr_maxblurpool_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(out), osb, osp, b, h, w, c,
                                                _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_maxblurpool_nhwc_fwd")
    return out


def instance_norm(x, eps=1e-5, leaky=None, inplace=False):
    """nn.InstanceNorm2d (affine=False) [+ LeakyReLU(leaky)] on channels-last data."""
    x = as_nhwc(x, "instance_norm input")
    b, c, h, w = x.shape
    out = x if inplace else empty_nhwc(b, c, h, w, x.device)
    if b == 0 or h * w == 0:
        return out
    lib = _lib.lib()
    nbytes = lib.sr_instance_norm_workspace_bytes(b, h, w, c)
    ws = _workspace(x.device, "inorm", nbytes)
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    with _lib.on_device(x.device):
        rc = lib.sr_instance_norm_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(out), osb, osp, b, h, w, c,
                                           C.c_float(eps), C.c_float(-1.0 if leaky is None else float(leaky)),
                                           _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_instance_norm_nhwc_fwd")
    return out


_PACKED_C16 = weakref.WeakKeyDictionary()  # nn.Conv2d -> (state key, packed weight, bias)


def instance_norm_stats(x, eps=1e-5):
    """Per-(image, channel) InstanceNorm statistics of a channels-last tensor: [B,2,C] = (mean, 1/sqrt(var + eps))."""
    x = as_nhwc(x, "instance_norm_stats input")
    b, c, h, w = x.shape
    stats = torch.empty((b, 2, c), dtype=torch.float32, device=x.device)
    if b == 0:
        return stats
    lib = _lib.lib()
    ws = _workspace(x.device, "inorm", lib.sr_instance_norm_workspace_bytes(b, h, w, c))
    isb, isp = _strides(x)
    with _lib.on_device(x.device):
        rc = lib.sr_instance_norm_stats_nhwc(_lib.ptr(x), isb, isp, b, h, w, c, C.c_float(eps), _lib.ptr(stats),
                                             _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_instance_norm_stats_nhwc")
    return stats


def conv1x1_stats(x, conv: nn.Conv2d, eps=1e-5):
    """(conv(x), InstanceNorm statistics [B,2,C] of conv(x)) for the matching encoder's Conv2d(64, 128, 1) ->
    InstanceNorm2d pair (reference networks.py:187-188): the statistics come out of the convolution's own pass."""
    _lib.refuse_autograd(x, conv.weight)
    x = as_nhwc(x, "conv input")
    b, ci, h, w = x.shape
    co = conv.out_channels
    if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or tuple(conv.padding) != (0, 0) or conv.groups != 1 \
            or (ci, co) != (64, 128) or ci != conv.in_channels:
        raise _lib.HipLibraryError(f"conv1x1_stats needs Conv2d(64, 128, 1), got {conv} for {ci} channels")
    out = empty_nhwc(b, co, h, w, x.device)
    stats = torch.empty((b, 2, co), dtype=torch.float32, device=x.device)
    if b == 0:
        return out, stats
    lib = _lib.lib()
    ws = _workspace(x.device, "c1s", lib.sr_conv1x1_stats_workspace_bytes(b, h, w, co))
    weight = conv.weight.detach().reshape(co, ci).contiguous()   # [128][64] rows, whatever the module's memory format
    bias = conv.bias.detach() if conv.bias is not None else None
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    prof = PROFILE
    with _lib.on_device(x.device):
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = lib.sr_conv1x1_stats_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), osb,
                                           osp, b, h, w, ci, co, C.c_float(eps), _lib.ptr(stats), _lib.ptr(ws),
                                           ws.numel() * 4, _lib.stream_ptr(x.device))
        if prof is not None:
            ev1.record()
            prof.append(("sr_conv1x1_stats_kernel", 2.0 * b * h * w * co * ci, ev0, ev1, (b, ci, h, w, co, 1, 1),
                         2.0 * b * ((h * w + 63) // 64) * 64 * co * ci))
    _lib.check(rc, "sr_conv1x1_stats_nhwc_fwd")
    return out, stats


def conv3x3_c16(x, conv: nn.Conv2d, in_stats=None, in_leaky=None, leaky=None):
    """conv3x3 (<= 16 output channels, zero or replicate padding) of act(InstanceNorm(x)) where the normalisation
    (statistics `in_stats` from instance_norm_stats) and its LeakyReLU are applied while the input is staged."""
    _lib.refuse_autograd(x, conv.weight)
    x = as_nhwc(x, "conv input")
    b, ci, h, w = x.shape
    co = conv.out_channels
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or tuple(conv.padding) != (1, 1) or conv.groups != 1 \
            or conv.dilation != (1, 1) or conv.padding_mode not in ("zeros", "replicate") or co > 16 or ci % 32 \
            or ci != conv.in_channels:
        raise _lib.HipLibraryError(f"conv3x3_c16 needs Conv2d(32k, <=16, 3, padding=1), got {conv} for {ci} channels")
    lib = _lib.lib()
    key = _state_key(conv, None)
    hit = _PACKED_C16.get(conv)
    if hit is None or hit[0] != key:
        wp = torch.empty(lib.sr_conv3x3_c16_packed_weight_floats(co, ci), dtype=torch.float32, device=conv.weight.device)
        with _lib.on_device(conv.weight.device):
            _lib.check(lib.sr_conv3x3_c16_pack_weights(_lib.ptr(conv.weight.detach().contiguous()), co, ci, _lib.ptr(wp),
                                                       _lib.stream_ptr(conv.weight.device)), "sr_conv3x3_c16_pack_weights")
        hit = (key, wp, conv.bias.detach() if conv.bias is not None else None)
        _packed_here(conv, "c16", conv.weight.device)
        _PACKED_C16[conv] = hit
    else:
        _await_packed(conv, "c16", conv.weight.device)
    _, wp, bias = hit
    if in_stats is not None and (tuple(in_stats.shape) != (b, 2, ci) or not in_stats.is_contiguous()):
        raise ValueError(f"in_stats must be a contiguous [{b}, 2, {ci}] tensor")
    out = empty_nhwc(b, co, h, w, x.device)
    if b == 0:
        return out
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    prof = PROFILE
    with _lib.on_device(x.device):
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = lib.sr_conv3x3_c16_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(in_stats),
                                         C.c_float(-1.0 if in_leaky is None else float(in_leaky)), _lib.ptr(wp),
                                         _lib.ptr(bias), _lib.ptr(out), osb, osp, b, h, w, ci, co,
                                         int(conv.padding_mode == "replicate"),
                                         C.c_float(-1.0 if leaky is None else float(leaky)), _lib.stream_ptr(x.device))
        if prof is not None:
            ev1.record()
            prof.append(("sr_t16_kernel", 2.0 * b * h * w * co * ci * 9, ev0, ev1, (b, ci, h, w, co, 3, 1),
                         2.0 * b * ((h + 7) // 8) * ((w + 15) // 16) * 128 * 16 * ci * 9))
    _lib.check(rc, "sr_conv3x3_c16_nhwc_fwd")
    return out


# ---- MBConv pieces of the image-prior encoder (csrc/sr_mbconv.hip) -----------------------------------------------

_PACKED_DW = weakref.WeakKeyDictionary()  # depthwise nn.Conv2d -> (state key, [9, C] weight, bias)


def packed_dw_weight(conv: nn.Conv2d, bn=None):
    """([9, C] tap-major depthwise weight with the eval-mode BatchNorm scale folded in, bias); cached."""
    _lib.require_device_f32("depthwise conv weight", conv.weight)
    c = conv.in_channels
    if conv.kernel_size != (3, 3) or conv.groups != c or conv.out_channels != c or conv.dilation != (1, 1) \
            or conv.padding_mode != "zeros":
        raise _lib.HipLibraryError(f"unsupported depthwise Conv2d configuration for the HIP path: {conv}")
    key = _state_key(conv, bn)
    hit = _PACKED_DW.get(conv)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    w, bias = _effective_weight(conv, bn)                      # [C, 1, 3, 3]
    w9c = w.reshape(c, 9).t().contiguous()
    bias = bias.contiguous() if bias is not None else None
    _PACKED_DW[conv] = (key, w9c, bias)
    return w9c, bias


def dwconv3x3(x, conv: nn.Conv2d, bn=None, leaky=None, act=None, tf_same=False, want_pool=False, pads=None):
    """act(bn(depthwise_conv3x3(x))) on a channels-last view; with want_pool also returns the per-band partial sums
    of the result ([B, bands, C]) from which `se_gate` finishes the squeeze-excite average pool."""
    _lib.refuse_autograd(x, conv.weight)
    x = as_nhwc(x, "depthwise conv input")
    b, c, h, w = x.shape
    if c != conv.in_channels:
        raise ValueError(f"depthwise conv expects {conv.in_channels} channels, got {c}")
    s = conv.stride[0]
    if conv.stride[0] != conv.stride[1] or s not in (1, 2):
        raise _lib.HipLibraryError(f"unsupported stride {conv.stride}")
    if pads is None and not tf_same and tuple(conv.padding) != (1, 1):
        raise _lib.HipLibraryError(f"unsupported padding {conv.padding}")
    if pads is None:
        pads = tf_same_pads(h, w, 3, s) if tf_same else (1, 1, 1, 1)
    ho, wo = (h + pads[0] + pads[2] - 3) // s + 1, (w + pads[1] + pads[3] - 3) // s + 1
    w9c, bias = packed_dw_weight(conv, bn)
    lib = _lib.lib()
    out = empty_nhwc(b, c, ho, wo, x.device)
    bands = lib.sr_dwconv3x3_pool_bands(ho)
    pool = torch.empty((b, bands, c), dtype=torch.float32, device=x.device) if want_pool else None
    if b > 0:
        isb, isp = _strides(x)
        osb, osp = _strides(out)
        with _lib.on_device(x.device):
            rc = lib.sr_dwconv3x3_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(w9c), _lib.ptr(bias), _lib.ptr(out), osb,
                                           osp, _lib.ptr(pool), b, h, w, c, s, *pads,
                                           C.c_float(_act_code(leaky, act)), _lib.stream_ptr(x.device))
        _lib.check(rc, "sr_dwconv3x3_nhwc_fwd")
    return (out, pool) if want_pool else out


def se_gate(pool_partial, pixels, conv_reduce: nn.Conv2d, conv_expand: nn.Conv2d):
    """sigmoid(conv_expand(silu(conv_reduce(mean)))) per image from dwconv3x3's partial sums: [B, C] gates."""
    _lib.require_device_f32("pool partial sums", pool_partial)
    b, bands, c = pool_partial.shape
    rd = conv_reduce.out_channels
    if conv_reduce.kernel_size != (1, 1) or conv_expand.kernel_size != (1, 1) or conv_reduce.in_channels != c \
            or conv_expand.in_channels != rd or conv_expand.out_channels != c:
        raise _lib.HipLibraryError("squeeze-excite expects 1x1 convs C -> rd -> C")
    _lib.refuse_autograd(pool_partial, conv_reduce.weight, conv_expand.weight)
    gate = torch.empty((b, c), dtype=torch.float32, device=pool_partial.device)
    if b == 0:
        return gate
    w1 = conv_reduce.weight.detach().reshape(rd, c)
    w2 = conv_expand.weight.detach().reshape(c, rd)
    b1 = conv_reduce.bias.detach() if conv_reduce.bias is not None else None
    b2 = conv_expand.bias.detach() if conv_expand.bias is not None else None
    for t in (w1, w2):
        _lib.require_device_f32("squeeze-excite weight", t)
    with _lib.on_device(gate.device):
        rc = _lib.lib().sr_se_gate_fwd(_lib.ptr(pool_partial.contiguous()), bands, pixels, _lib.ptr(w1.contiguous()),
                                       _lib.ptr(b1), _lib.ptr(w2.contiguous()), _lib.ptr(b2), _lib.ptr(gate), b, c, rd,
                                       _lib.stream_ptr(gate.device))
    _lib.check(rc, "sr_se_gate_fwd")
    return gate


_PACKED_RGB = weakref.WeakKeyDictionary()  # stem nn.Conv2d -> (state key, [27, Cout] weight with BN folded, bias)


def rgb_stem3x3s2(image, conv: nn.Conv2d, bn=None, act=None, leaky=None, tf_same=True):
    """act(bn(conv3x3_s2(image))) for a 3-channel image (EfficientNetV2's conv_stem): a VALU kernel that reads the image
    through its strides; returns a channels-last [B, Cout, Ho, Wo] tensor.  Falls back to conv2d for other layers."""
    _lib.require_device_f32("image", image)
    _lib.refuse_autograd(image, conv.weight)
    if conv.kernel_size != (3, 3) or conv.stride != (2, 2) or conv.in_channels != 3 or conv.groups != 1 or \
            conv.out_channels != 24 or image.dim() != 4 or image.shape[1] != 3:
        return conv2d(image, conv, bn=bn, act=act, leaky=leaky, tf_same=tf_same)
    b, _, h, w = image.shape
    pads = tf_same_pads(h, w, 3, 2) if tf_same else (1, 1, 1, 1)
    ho, wo = (h + pads[0] + pads[2] - 3) // 2 + 1, (w + pads[1] + pads[3] - 3) // 2 + 1
    key = _state_key(conv, bn)
    hit = _PACKED_RGB.get(conv)
    if hit is None or hit[0] != key:
        wt, bias = _effective_weight(conv, bn)                          # [Cout, 3, 3, 3]
        w27 = wt.permute(2, 3, 1, 0).reshape(27, conv.out_channels).contiguous()   # [ky][kx][ci][Cout]
        hit = (key, w27, bias.contiguous() if bias is not None else None)
        _packed_here(conv, "rgb", conv.weight.device)
        _PACKED_RGB[conv] = hit
    else:
        _await_packed(conv, "rgb", conv.weight.device)
    _, w27, bias = hit
    out = empty_nhwc(b, conv.out_channels, ho, wo, image.device)
    if b == 0:
        return out
    sb, sc, sy, sx = image.stride()
    osb, osp = _strides(out)
    with _lib.on_device(image.device):
        rc = _lib.lib().sr_rgb_stem3x3s2_fwd(_lib.ptr(image), sb, sc, sy, sx, _lib.ptr(w27), _lib.ptr(bias), _lib.ptr(out), osb,
                                             osp, b, h, w, conv.out_channels, pads[0], pads[1], ho, wo,
                                             C.c_float(_act_code(leaky, act)), _lib.stream_ptr(image.device))
    _lib.check(rc, "sr_rgb_stem3x3s2_fwd")
    return out


def se_gates(pool_partial, pixels, conv_reduce: nn.Conv2d, conv_expand: nn.Conv2d):
    """[B, C] squeeze-excite gates from dwconv3x3's partial sums (two short launches); the consumer -- conv2d(..., gate=) --
    applies them to its input while loading it, so the gated map is never written."""
    _lib.require_device_f32("pool partial sums", pool_partial)
    b, bands, c = pool_partial.shape
    rd = conv_reduce.out_channels
    if conv_reduce.kernel_size != (1, 1) or conv_expand.kernel_size != (1, 1) or conv_reduce.in_channels != c \
            or conv_expand.in_channels != rd or conv_expand.out_channels != c or not pool_partial.is_contiguous():
        raise _lib.HipLibraryError("squeeze-excite expects 1x1 convs C -> rd -> C and contiguous [B, bands, C] partial sums")
    _lib.refuse_autograd(pool_partial, conv_reduce.weight, conv_expand.weight)
    gate = torch.empty((b, c), dtype=torch.float32, device=pool_partial.device)
    if b == 0:
        return gate
    hidden = torch.empty((b, rd), dtype=torch.float32, device=pool_partial.device)
    w1, w2 = conv_reduce.weight.detach(), conv_expand.weight.detach()
    if not (w1.is_contiguous() and w2.is_contiguous()):
        w1, w2 = w1.contiguous(), w2.contiguous()
    b1 = conv_reduce.bias.detach() if conv_reduce.bias is not None else None
    b2 = conv_expand.bias.detach() if conv_expand.bias is not None else None
    with _lib.on_device(gate.device):
        rc = _lib.lib().sr_se_gate2_fwd(_lib.ptr(pool_partial), bands, pixels, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2),
                                        _lib.ptr(b2), _lib.ptr(hidden), _lib.ptr(gate), b, c, rd,
                                        _lib.stream_ptr(gate.device))
    _lib.check(rc, "sr_se_gate2_fwd")
    return gate


# (device, stream) -> zeroed int32 arrival counters of sr_mbconv_expand_dw_se_fwd (the call leaves them zeroed).  Per STREAM: two
# fused launches in flight on different streams must not interleave their atomicAdds on one buffer -- launches on one
# stream are ordered.  A captured graph keeps the buffer of the stream it was captured on.
_MBX_COUNTERS = {}


def mbconv_fused_supported(x, conv_pw: nn.Conv2d, conv_dw: nn.Conv2d, se):
    """True when the one-launch front half (csrc/sr_mbconv_fused.hip) serves this MBConv block on this input."""
    if conv_dw.stride != (1, 1) or tuple(conv_dw.padding) != (1, 1) or conv_pw.kernel_size != (1, 1):
        return False
    return bool(_shape_query(_lib.lib(), "sr_mbconv_fused_supported", x.shape[2], x.shape[3], conv_pw.in_channels,
                             conv_pw.out_channels, se.conv_reduce.out_channels))


def mbconv_expand_dw_se(x, conv_pw: nn.Conv2d, bn1, conv_dw: nn.Conv2d, bn2, se):
    """MBConv front half in one launch: silu(bn1(conv_pw(x))) -> silu(bn2(conv_dw(.))) -> squeeze-excite gates.  Returns
    (the depthwise output as a channels-last view, gates [B, mid]); the projection applies the gates (conv2d(..., gate=))."""
    _lib.refuse_autograd(x, conv_pw.weight)
    x = as_nhwc(x, "MBConv input")
    b, ci, h, w = x.shape
    mid, rd = conv_pw.out_channels, se.conv_reduce.out_channels
    w_exp, b_exp = gemm_weight(conv_pw, bn1)
    w9c, b_dw = packed_dw_weight(conv_dw, bn2)
    w1, w2 = se.conv_reduce.weight.detach(), se.conv_expand.weight.detach()
    if not (w1.is_contiguous() and w2.is_contiguous()):
        w1, w2 = w1.contiguous(), w2.contiguous()
    b1 = se.conv_reduce.bias.detach() if se.conv_reduce.bias is not None else None
    b2 = se.conv_expand.bias.detach() if se.conv_expand.bias is not None else None
    out = empty_nhwc(b, mid, h, w, x.device)
    pool = torch.empty((b, mid), dtype=torch.float32, device=x.device)
    gate = torch.empty((b, mid), dtype=torch.float32, device=x.device)
    if b == 0:
        return out, gate
    ckey = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    cnt = _MBX_COUNTERS.get(ckey)
    if cnt is None or cnt.numel() < b:
        cnt = _MBX_COUNTERS[ckey] = torch.zeros(max(b, 64), dtype=torch.int32, device=x.device)
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    with _lib.on_device(x.device):
        rc = _lib.lib().sr_mbconv_expand_dw_se_fwd(_lib.ptr(x), isb, isp, _lib.ptr(w_exp), _lib.ptr(b_exp), _lib.ptr(w9c),
                                                   _lib.ptr(b_dw), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2),
                                                   _lib.ptr(out), osb, osp, _lib.ptr(pool), _lib.ptr(gate), _lib.ptr(cnt), b, h, w,
                                                   ci, mid, rd, _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_mbconv_expand_dw_se_fwd")
    return out, gate


def se_scale_(x, pool_partial, conv_reduce: nn.Conv2d, conv_expand: nn.Conv2d, want_gate=False):
    """x *= squeeze-excite gate, in place on a channels-last view (se_gate + scale_channels_ in two short launches)."""
    _lib.require_device_f32("pool partial sums", pool_partial)
    if not _is_nhwc_view(x):
        raise ValueError("se_scale_ needs a channels-last view")
    b, c, h, w = x.shape
    rd = conv_reduce.out_channels
    if conv_reduce.kernel_size != (1, 1) or conv_expand.kernel_size != (1, 1) or conv_reduce.in_channels != c \
            or conv_expand.in_channels != rd or conv_expand.out_channels != c or pool_partial.shape[0] != b \
            or pool_partial.shape[2] != c or not pool_partial.is_contiguous():
        raise _lib.HipLibraryError("squeeze-excite expects 1x1 convs C -> rd -> C and [B, bands, C] partial sums")
    _lib.refuse_autograd(x, pool_partial, conv_reduce.weight, conv_expand.weight)
    gate = torch.empty((b, c), dtype=torch.float32, device=x.device) if want_gate else None
    if b == 0:
        return (x, gate) if want_gate else x
    hidden = torch.empty((b, rd), dtype=torch.float32, device=x.device)
    w1, w2 = conv_reduce.weight.detach(), conv_expand.weight.detach()
    if not (w1.is_contiguous() and w2.is_contiguous()):
        w1, w2 = w1.contiguous(), w2.contiguous()
    b1 = conv_reduce.bias.detach() if conv_reduce.bias is not None else None
    b2 = conv_expand.bias.detach() if conv_expand.bias is not None else None
    sb, sp = _strides(x)
    with _lib.on_device(x.device):
        rc = _lib.lib().sr_se_scale_nhwc_fwd(_lib.ptr(pool_partial), pool_partial.shape[1], _lib.ptr(w1), _lib.ptr(b1),
                                             _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(hidden), _lib.ptr(x), sb, sp,
                                             _lib.ptr(x), sb, sp, _lib.ptr(gate), b, h, w, c, rd,
                                             _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_se_scale_nhwc_fwd")
    return (x, gate) if want_gate else x


def scale_channels_(x, gate):
    """x[b, c, :, :] *= gate[b, c] in place on a channels-last view."""
    _lib.require_device_f32("gate", gate)
    if not _is_nhwc_view(x):
        raise ValueError("scale_channels_ needs a channels-last view")
    _lib.refuse_autograd(x, gate)
    b, c, h, w = x.shape
    if tuple(gate.shape) != (b, c) or not gate.is_contiguous():
        raise ValueError(f"gate must be a contiguous [{b}, {c}] tensor")
    if b == 0:
        return x
    sb, sp = _strides(x)
    with _lib.on_device(x.device):
        rc = _lib.lib().sr_scale_channels_nhwc_fwd(_lib.ptr(x), sb, sp, _lib.ptr(gate), _lib.ptr(x), sb, sp, b, h, w, c,
                                                   _lib.stream_ptr(x.device))
    _lib.check(rc, "sr_scale_channels_nhwc_fwd")
    return x


def add_(y, x):
    """y += x on channels-last views (same shape)."""
    _lib.require_device_f32("y", y)
    x = as_nhwc(x, "x")
    if not _is_nhwc_view(y) or tuple(x.shape) != tuple(y.shape):
        raise ValueError("add_ needs two channels-last views of the same shape")
    _lib.refuse_autograd(x, y)
    b, c, h, w = y.shape
    if b == 0:
        return y
    ysb, ysp = _strides(y)
    xsb, xsp = _strides(x)
    with _lib.on_device(y.device):
        rc = _lib.lib().sr_add_nhwc_fwd(_lib.ptr(y), ysb, ysp, _lib.ptr(x), xsb, xsp, _lib.ptr(y), ysb, ysp, b, h, w, c,
                                        _lib.stream_ptr(y.device))
    _lib.check(rc, "sr_add_nhwc_fwd")
    return y
