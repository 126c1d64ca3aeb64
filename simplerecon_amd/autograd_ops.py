"""Differentiable conv-stack operators: the training path of BasicBlock / CVEncoder / DepthDecoderPP (reference
train.py:126-145 runs autograd through modules/layers.py:24-85 and modules/networks.py:20-127).

torch.autograd is the tape (plumbing); every arithmetic step, forward and backward, is a HIP kernel:
  forward        the same MFMA kernels as inference (direct / Winograd), weights packed per call (they change every step);
  data gradient  the forward kernels on the flipped, transposed weight (stride 2: on the zero-stuffed output gradient);
  weight / bias  csrc/sr_conv_bwd.hip (MFMA split over pixels, partial slabs + ordered reduce: deterministic);
  LeakyReLU      from the saved output; bilinear x2: its adjoint kernel.
Gradients are pinned to the reference's own autograd (tests/golden/grad_block_*.npz, grad_cv_encoder_narrow.npz,
grad_decoder_narrow.npz).  Not covered yet: the two encoders (no BatchNorm / InstanceNorm backward) -- DepthModel treats
their outputs as constants."""
import ctypes as C
import os

import torch
from torch import nn

from . import _lib
from .ops import _is_nhwc_view, _strides, _workspace, as_nhwc, empty_nhwc


# torch.autocast compatibility (the reference trains under 16-bit autocast, options.py:100-101, train.py:132): every
# differentiable operator here runs its fp32 HIP kernels whatever the autocast state -- floating-point CUDA inputs that
# arrive in fp16 / bf16 are cast to fp32 at the operator's entry (a differentiable cast: gradients go back in the caller's
# dtype) and autocast is off inside.  fp32 compute and fp32 storage: more accurate than the reference's half-precision
# convolutions, without their memory saving (an fp16-storage variant of the kernels is not built).
_amp_cast_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def _amp_fwd(fwd):
    """custom_fwd(cast_inputs=fp32) that also records on ctx whether the caller was inside an autocast region (and its
    dtype): `_stash` then keeps the saved activations in that 16-bit dtype."""
    cast = _amp_cast_fwd(fwd)

    def wrapper(ctx, *args, **kwargs):
        on = torch.is_autocast_enabled("cuda")
        ctx._sr_autocast = on
        ctx._sr_autocast_dtype = torch.get_autocast_dtype("cuda") if on else None
        return cast(ctx, *args, **kwargs)
    wrapper.__name__, wrapper.__doc__ = getattr(fwd, "__name__", "forward"), fwd.__doc__
    return wrapper


# fp16 / bf16 STORAGE of the saved activations under torch.autocast (SR_AUTOCAST_HALF_STORAGE=1; default since r06: fp32).  The
# reference's autocast keeps activations in half precision (options.py:100-101); here the kernels compute and hand over
# fp32, but what autograd SAVES for the backward pass -- the bulk of a training step's memory -- is stored in the autocast
# dtype and widened again when the backward kernel needs it.  A tensor saved by several operators (a producer saves its
# output for the activation's derivative, the consumer saves it as its input) is narrowed once and shared.
#
# r06: both 16-bit switches are OFF by default.  They buy memory (3.8 instead of 5.9 GiB peak at batch 2, 640x480, 7 views) and
# cost time: the backward kernels are fp32, so every saved tensor is narrowed once and widened once by cast kernels (892 + 593
# launches per step).  Measured training step (scripts/train_step_micro.py 2, profiles/r06_train_autocast.txt): fp32 77.7 ms;
# bf16 autocast with storage + I/O 87.0, storage only 85.8, I/O only 83.1, neither (fp32 kernels under autocast) 78.4.  Until the
# backward kernels read 16-bit operands themselves, "autocast is not slower than fp32"  is the default and the
# memory saving is opt-in: SR_AUTOCAST_HALF_STORAGE=1 SR_AUTOCAST_HALF_IO=1.
STORE_HALF = os.environ.get("SR_AUTOCAST_HALF_STORAGE", "0") != "0"
# 16-bit KERNEL I/O under torch.autocast (r04; SR_AUTOCAST_HALF_IO=1; default since r06: fp32 kernels).  Inside an
# autocast region the convolutions of the conv stack (BasicBlock / CVEncoder / DepthDecoderPP: _ConvBiasAct) read and write
# their activations in the autocast dtype -- the Winograd and pointwise kernels load four fp16 / bf16 channels as one 8-byte
# access, widen them on the way into LDS / registers, accumulate in fp32 (fp32 weights, fp32 MFMA) and round the result once
# on the way out (sr_conv3x3_wino_io_nhwc_fwd, sr_pw_conv_io_nhwc_fwd); what flows between the layers and what autograd saves
# is 16-bit, like the reference's `precision: 16` training (options.py:100-101, train.py:132), with fp32 instead of fp16
# accumulation inside a layer.  The backward kernels (weight / bias gradients, activation derivative, data gradient) run on
# fp32 copies of the saved tensors; stride-2 and explicitly padded convolutions convert at their boundary.
HALF_IO = os.environ.get("SR_AUTOCAST_HALF_IO", "0") != "0"
_IO_CODE = {torch.float16: 1, torch.bfloat16: 2}


def _amp_state_fwd(fwd):
    """Like `_amp_fwd` but WITHOUT casting the inputs: records the caller's autocast state on ctx and runs the body with
    autocast off; the operator decides what dtype its kernels read and write."""
    def wrapper(ctx, *args, **kwargs):
        on = torch.is_autocast_enabled("cuda")
        ctx._sr_autocast = on
        ctx._sr_autocast_dtype = torch.get_autocast_dtype("cuda") if on else None
        with torch.autocast(device_type="cuda", enabled=False):
            return fwd(ctx, *args, **kwargs)
    wrapper.__name__, wrapper.__doc__ = getattr(fwd, "__name__", "forward"), fwd.__doc__
    return wrapper


def _amp_state_bwd(bwd):
    def wrapper(ctx, *grads):
        with torch.autocast(device_type="cuda", enabled=False):
            return bwd(ctx, *grads)
    wrapper.__name__, wrapper.__doc__ = getattr(bwd, "__name__", "backward"), bwd.__doc__
    return wrapper


def _nhwc_any(t, name):
    """A channels-last view of a CUDA tensor of any of the I/O dtypes (fp32 / fp16 / bf16)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise _lib.HipLibraryError(f"{name}: expected a CUDA fp32 / fp16 / bf16 tensor, got {getattr(t, 'dtype', type(t))} on "
                                   f"{getattr(t, 'device', '?')}")
    if t.dim() != 4:
        raise ValueError(f"{name} must be [B,C,H,W], got {tuple(t.shape)}")
    return t if _is_nhwc_view(t) else t.contiguous(memory_format=torch.channels_last)


def _stash(ctx, *tensors):
    half = STORE_HALF and getattr(ctx, "_sr_autocast", False)
    dt = getattr(ctx, "_sr_autocast_dtype", None)
    keep, flags = [], []
    for t in tensors:
        narrow = half and isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.dim() == 4 and \
            not isinstance(t, nn.Parameter) and t.numel() >= 4096
        if narrow:
            # the 16-bit copy is shared between the operators that save this tensor; it is valid only for the very
            # contents it was made from (an input refreshed in place with copy_ must be narrowed again)
            rec = getattr(t, "_sr_narrow", None)
            if rec is None or rec[0] != t._version or rec[1] != t.data_ptr() or rec[2].dtype != dt:
                rec = (t._version, t.data_ptr(), t.detach().to(dt))
                t._sr_narrow = rec
            keep.append(rec[2])
        else:
            keep.append(t)
        flags.append(narrow)
    ctx.save_for_backward(*keep)
    ctx._sr_stash_flags = flags


def _unstash(ctx):
    return tuple(t.float() if (f and t is not None) else t for t, f in zip(ctx.saved_tensors, ctx._sr_stash_flags))

FUSED_ACT_BIAS = os.environ.get("SR_FUSED_ACT_BIAS", "1") != "0"   # 0: separate sr_act_bwd + sr_bias_grad_nhwc launches (r02 a/b)


def grad_wanted(*tensors_or_modules):
    """True when autograd is recording and any tensor / module parameter asks for a gradient."""
    if not torch.is_grad_enabled():
        return False
    for t in tensors_or_modules:
        if isinstance(t, torch.Tensor):
            if t.requires_grad:
                return True
        elif isinstance(t, nn.Module):
            if any(p.requires_grad for p in t.parameters()):
                return True
        elif isinstance(t, (list, tuple)):
            if grad_wanted(*t):
                return True
    return False


# Every registration of a submodule anywhere in the process (add_module / attribute assignment: SyncBatchNorm conversion,
# BatchNorm fusion, a swapped block) bumps this counter through torch's global registration hook; cached layer lists carry the
# value they were gathered under.
_TREE_EPOCH = [0]


def _bump_tree_epoch(module, name, submodule):
    _TREE_EPOCH[0] += 1


torch.nn.modules.module.register_module_module_registration_hook(_bump_tree_epoch)


def any_batchnorm_training(module):
    """True if a batch-norm layer of `module` (BatchNorm2d, SyncBatchNorm, ... -- any `_BatchNorm`) is in training mode (batch
    statistics cannot be folded into the conv weights).  The layer list is gathered once per module TREE: walking `modules()`
    costs ~1 ms per forward on the EfficientNetV2-S pyramid, so the list is cached together with the process-wide submodule
    registration count it was gathered under -- replacing a layer anywhere (also deep in the tree, which r04's key of
    (training, number of direct children) missed) invalidates it, and every cached layer is re-checked against its parent's
    `_modules` entry (mutations the hook cannot see).  The layers' own `training` flags are read on every call."""
    rec = module.__dict__.get("_sr_bn_layers")
    if rec is not None and rec[0] == _TREE_EPOCH[0]:
        # the registration hook does not see direct mutations of `_modules` (del m.bn, ModuleList.__delitem__ / insert, fx or
        # quantisation swaps): every cached layer must still sit where it was found (~100 dict lookups)
        for parent, name, layer in rec[1]:
            if parent._modules.get(name) is not layer:
                rec = None
                break
    if rec is None or rec[0] != _TREE_EPOCH[0]:
        found = []
        for parent in module.modules():
            for name, child in parent._modules.items():
                if isinstance(child, nn.modules.batchnorm._BatchNorm):
                    found.append((parent, name, child))
        if isinstance(module, nn.modules.batchnorm._BatchNorm):
            found.append((_Holder(module), "self", module))
        rec = (_TREE_EPOCH[0], found)
        module.__dict__["_sr_bn_layers"] = rec
    for _, _, m in rec[1]:
        if m.training:
            return True
    return False


class _Holder:
    """Stand-in parent for a batch-norm layer that is itself the root of the query."""
    def __init__(self, m):
        self._modules = {"self": m}


def _dense_nhwc(t):
    """Channels-last, dense (pixel stride = C): what the elementwise backward kernels and the wgrad staging expect."""
    t = as_nhwc(t, "gradient")
    c, h, w = t.shape[1], t.shape[2], t.shape[3]
    if t.stride(1) != 1 or t.stride(3) != c or t.stride(2) != w * c or (t.shape[0] > 1 and t.stride(0) != h * w * c):
        t = t.contiguous(memory_format=torch.channels_last)
        if t.shape[0] > 1 and t.stride(0) != h * w * c:   # (contiguous() keeps a batch-strided view whose images are dense)
            t = t.clone(memory_format=torch.channels_last)
    return t


def _conv_raw_io(x, weight, bias, stride, residual, slope, pads):
    """_conv_raw on fp16 / bf16 activations: 16-bit kernel I/O where the kernel has it (3x3 / stride 1 through Winograd,
    1x1 / stride 1 through the pointwise GEMM), boundary conversion around the fp32 kernel otherwise."""
    lib = _lib.lib()
    dt = x.dtype
    x = _nhwc_any(x, "conv input")
    b, ci, h, w = x.shape
    co, ci_w, k, _ = weight.shape
    if ci_w != ci:
        raise ValueError(f"conv weight expects {ci_w} input channels, got {ci}")
    if residual is not None:
        residual = _nhwc_any(residual if residual.dtype == dt else residual.to(dt), "residual")
    wino = pads is None and stride == 1 and k == 3 and b > 0 and ci % 4 == 0 and co % 4 == 0 and \
        bool(lib.sr_conv_prefers_wino(b, h, w, ci, co, k, stride))
    pw = pads is None and stride == 1 and k == 1 and b > 0 and ci % 4 == 0
    aligned = x.data_ptr() % 8 == 0 and _strides(x)[1] % 4 == 0 and _strides(x)[0] % 4 == 0 and \
        (residual is None or (residual.data_ptr() % 8 == 0 and _strides(residual)[1] % 4 == 0 and _strides(residual)[0] % 4 == 0))
    if not ((wino or pw) and aligned):
        y = _conv_raw(x.float(), weight, bias, stride, residual.float() if residual is not None else None, slope, pads)
        return y.to(dt)
    out = torch.empty((b, co, h, w), dtype=dt, device=x.device, memory_format=torch.channels_last)
    wd = weight.detach().float().contiguous()
    st = _lib.stream_ptr(x.device)
    isb, isp = _strides(x)
    osb, osp = _strides(out)
    rsb, rsp = _strides(residual) if residual is not None else (0, 0)
    bd = bias.detach().float().contiguous() if bias is not None else None
    sl = C.c_float(-1.0 if slope is None else float(slope))
    with _lib.on_device(x.device):
        if wino:
            wp = torch.empty(lib.sr_wino_packed_weight_floats(co, ci), dtype=torch.float32, device=x.device)
            _lib.check(lib.sr_wino_pack_weights(_lib.ptr(wd), co, ci, _lib.ptr(wp), st), "sr_wino_pack_weights")
            rc = lib.sr_conv3x3_wino_io_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bd), _lib.ptr(residual), rsb, rsp,
                                                 _lib.ptr(out), osb, osp, b, h, w, ci, co, sl, _IO_CODE[dt], st)
        else:
            wp = torch.empty(lib.sr_conv_packed_weight_floats(co, ci, 1), dtype=torch.float32, device=x.device)
            _lib.check(lib.sr_conv_pack_weights(_lib.ptr(wd), co, ci, 1, _lib.ptr(wp), st), "sr_conv_pack_weights")
            rc = lib.sr_pw_conv_io_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bd), _lib.ptr(residual), rsb, rsp,
                                            _lib.ptr(out), osb, osp, b, h * w, ci, co, sl, _IO_CODE[dt], st)
    _lib.check(rc, "conv forward (16-bit I/O)")
    return out


def _conv_raw(x, weight, bias, stride, residual=None, slope=None, pads=None):
    """act(conv(x, weight) + bias [+ residual]) on the inference kernels with a weight TENSOR [Co, Ci, k, k]
    (packed on the fly; padding k // 2, or explicit (top, left, bottom, right) zero `pads`)."""
    if x.dtype in _IO_CODE:
        return _conv_raw_io(x, weight, bias, stride, residual, slope, pads)
    lib = _lib.lib()
    x = as_nhwc(x, "conv input")
    b, ci, h, w = x.shape
    co, ci_w, k, _ = weight.shape
    if ci_w != ci:
        raise ValueError(f"conv weight expects {ci_w} input channels, got {ci}")
    pt, pl, pb, pr = pads if pads is not None else (k // 2,) * 4
    ho, wo = (h + pt + pb - k) // stride + 1, (w + pl + pr - k) // stride + 1
    out = empty_nhwc(b, co, ho, wo, x.device)
    if b == 0:
        return out
    wd = weight.detach().contiguous()
    use_wino = pads is None and stride == 1 and k == 3 and bool(lib.sr_conv_prefers_wino(b, h, w, ci, co, k, stride))
    st = _lib.stream_ptr(x.device)
    with _lib.on_device(x.device):
        if use_wino:
            wp = torch.empty(lib.sr_wino_packed_weight_floats(co, ci), dtype=torch.float32, device=x.device)
            _lib.check(lib.sr_wino_pack_weights(_lib.ptr(wd), co, ci, _lib.ptr(wp), st), "sr_wino_pack_weights")
        else:
            wp = torch.empty(lib.sr_conv_packed_weight_floats(co, ci, k), dtype=torch.float32, device=x.device)
            _lib.check(lib.sr_conv_pack_weights(_lib.ptr(wd), co, ci, k, _lib.ptr(wp), st), "sr_conv_pack_weights")
        isb, isp = _strides(x)
        osb, osp = _strides(out)
        if residual is not None:
            residual = as_nhwc(residual, "residual")
        rsb, rsp = _strides(residual) if residual is not None else (0, 0)
        bd = bias.detach().contiguous() if bias is not None else None
        sl = C.c_float(-1.0 if slope is None else float(slope))
        if use_wino:
            rc = lib.sr_conv3x3_wino_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bd), _lib.ptr(residual), rsb,
                                              rsp, _lib.ptr(out), osb, osp, b, h, w, ci, co, sl, st)
        elif pads is not None:
            rc = lib.sr_conv2d_padded_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bd), _lib.ptr(residual), rsb,
                                               rsp, _lib.ptr(out), osb, osp, b, h, w, ci, co, k, stride, pt, pl, pb, pr, sl,
                                               st)
        else:
            rc = lib.sr_conv2d_nhwc_fwd(_lib.ptr(x), isb, isp, _lib.ptr(wp), _lib.ptr(bd), _lib.ptr(residual), rsb, rsp,
                                        _lib.ptr(out), osb, osp, b, h, w, ci, co, k, stride, sl, st)
    _lib.check(rc, "conv forward")
    return out


class _ConvBiasAct(torch.autograd.Function):
    """y = act(conv(x, W, stride, pad = k // 2 or explicit `pads`) + b [+ residual]), act = LeakyReLU(slope) or identity
    (slope None).  `pads` = (top, left, bottom, right) zero padding (TF-"SAME" convolutions of the image-prior encoder,
    the valid convolution behind the matching encoder's replicate pad)."""

    @staticmethod
    @_amp_state_fwd
    def forward(ctx, x, weight, bias, residual, stride, slope, pads=None):
        # activation dtype of this layer: the autocast dtype with 16-bit kernel I/O, fp32 otherwise (inputs that arrive
        # in another floating dtype are cast: a differentiable cast, gradients go back in the caller's dtype)
        dt = ctx._sr_autocast_dtype if (ctx._sr_autocast and HALF_IO and ctx._sr_autocast_dtype in _IO_CODE) else torch.float32
        if not ctx._sr_autocast:   # outside autocast the path is fp32 only, loudly (as everywhere in this package)
            for name, t in (("conv input", x), ("conv weight", weight)) + ((("residual", residual),) if residual is not None else ()):
                _lib.require_device_f32(name, t)
        ctx.in_dtypes = (x.dtype, residual.dtype if residual is not None else None)
        x = x if x.dtype == dt else x.to(dt)
        if residual is not None and residual.dtype != dt:
            residual = residual.to(dt)
        weight = weight if weight.dtype == torch.float32 else weight.float()
        if bias is not None and bias.dtype != torch.float32:
            bias = bias.float()
        if dt == torch.float32:
            for name, t in (("conv input", x), ("conv weight", weight)):
                _lib.require_device_f32(name, t)
            x = as_nhwc(x, "conv input")
        out = _conv_raw(x, weight, bias, stride, residual, slope, pads)
        ctx.stride, ctx.slope, ctx.pads = stride, slope, pads
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        _stash(ctx, x, weight, out if slope is not None else None)
        return out

    @staticmethod
    @_amp_state_bwd
    def backward(ctx, g):
        x, weight, out = _unstash(ctx)
        # (16-bit I/O: the backward kernels take fp32 copies of what the forward saved in 16 bits)
        x = x if x.dtype == torch.float32 else x.float()
        out = out if (out is None or out.dtype == torch.float32) else out.float()
        lib = _lib.lib()
        dev = x.device
        b, ci, h, w = x.shape
        co, _, k, _ = weight.shape
        s = ctx.stride
        ho, wo = g.shape[2], g.shape[3]
        g = _dense_nhwc(g if g.dtype == torch.float32 else g.float())
        st = _lib.stream_ptr(dev)
        need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
        want_b = ctx.has_bias and need_b
        fused = FUSED_ACT_BIAS and co % 4 == 0 and b > 0 and (ctx.slope is not None or want_b)
        d_b = None
        with _lib.on_device(dev):
            if fused:
                # one float4 pass: LeakyReLU' of the saved output and / or the bias gradient (deterministic two-stage sum)
                gp = torch.empty_like(g) if ctx.slope is not None else g
                px = b * ho * wo
                nws = lib.sr_act_bwd_bias_workspace_bytes(px, co) if want_b else 0
                ws = _workspace(dev, "bias_grad", nws) if want_b else None
                if want_b:
                    d_b = torch.empty((co,), dtype=torch.float32, device=dev)
                _lib.check(lib.sr_act_bwd_bias_nhwc(_lib.ptr(g), _lib.ptr(_dense_nhwc(out)) if ctx.slope is not None else None,
                                                    _lib.ptr(gp) if ctx.slope is not None else None, _lib.ptr(d_b), px, co,
                                                    C.c_float(float(ctx.slope) if ctx.slope is not None else 0.0),
                                                    _lib.ptr(ws), nws, st), "sr_act_bwd_bias_nhwc")
            elif ctx.slope is not None:
                gp = torch.empty_like(g)
                _lib.check(lib.sr_act_bwd(_lib.ptr(g), _lib.ptr(_dense_nhwc(out)), _lib.ptr(gp), g.numel(),
                                          C.c_float(float(ctx.slope)), st), "sr_act_bwd")
            else:
                gp = g
            d_x = d_w = None
            gsb, gsp = _strides(gp)
            pads = ctx.pads
            if need_w and b > 0:
                # dense [Co][Ci][k][k]: what sr_conv_wgrad_nhwc writes -- NOT empty_like (a channels_last weight's strides)
                d_w = torch.empty(weight.shape, dtype=torch.float32, device=dev)
                xsb, xsp = _strides(x)
                if pads is None:
                    nws = lib.sr_conv_wgrad_workspace_bytes(b, h, w, ci, co, k, s)
                    ws = _workspace(dev, "wgrad", nws)
                    _lib.check(lib.sr_conv_wgrad_nhwc(_lib.ptr(x), xsb, xsp, _lib.ptr(gp), gsb, gsp, _lib.ptr(d_w), b, h, w,
                                                      ci, co, k, s, _lib.ptr(ws), nws, st), "sr_conv_wgrad_nhwc")
                else:
                    nws = lib.sr_conv_wgrad_padded_workspace_bytes(b, ho, wo, ci, co, k)
                    ws = _workspace(dev, "wgrad", nws)
                    _lib.check(lib.sr_conv_wgrad_padded_nhwc(_lib.ptr(x), xsb, xsp, _lib.ptr(gp), gsb, gsp, _lib.ptr(d_w), b,
                                                             h, w, ci, co, k, s, pads[0], pads[1], ho, wo, _lib.ptr(ws), nws,
                                                             st), "sr_conv_wgrad_padded_nhwc")
            elif need_w:
                d_w = torch.zeros(weight.shape, dtype=torch.float32, device=dev)
            if want_b and d_b is None:
                d_b = torch.empty((co,), dtype=torch.float32, device=dev)
                _lib.check(lib.sr_bias_grad_nhwc(_lib.ptr(gp), gsb, gsp, _lib.ptr(d_b), b, ho, wo, co, st),
                           "sr_bias_grad_nhwc")
            if need_x:
                wt = torch.empty((ci, co, k, k), dtype=torch.float32, device=dev)
                _lib.check(lib.sr_conv_flip_transpose_weights(_lib.ptr(weight.detach().contiguous()), co, ci, k,
                                                              _lib.ptr(wt), st), "sr_conv_flip_transpose_weights")
                if s == 1:
                    src = gp
                else:  # stride 2: dL/dx = conv_s1(zero-stuffed dL/dy, flip(W)^T) on the input's grid
                    src = empty_nhwc(b, co, h, w, dev)
                    if b > 0:
                        _lib.check(lib.sr_zero_stuff2x_nhwc(_lib.ptr(gp), gsb, gsp, _lib.ptr(src), b, ho, wo, h, w, co, st),
                                   "sr_zero_stuff2x_nhwc")
                if pads is None:
                    d_x = _conv_raw(src, wt, None, 1)
                elif s == 1:   # full correlation: pads k-1-p on the opposite roles
                    d_x = _conv_raw(src, wt, None, 1, pads=(k - 1 - pads[0], k - 1 - pads[1], k - 1 - pads[2], k - 1 - pads[3]))
                else:          # the stuffed gradient lives on the input's H x W grid: output H x W needs (k-1-pt, k-1-pl, pt, pl)
                    d_x = _conv_raw(src, wt, None, 1, pads=(k - 1 - pads[0], k - 1 - pads[1], pads[0], pads[1]))
        d_r = gp if (ctx.has_res and need_r) else None
        xdt, rdt = ctx.in_dtypes
        if d_x is not None and d_x.dtype != xdt:
            d_x = d_x.to(xdt)
        if d_r is not None and d_r.dtype != rdt:
            d_r = d_r.to(rdt)
        return d_x, d_w, d_b, d_r, None, None, None


class _Upsample2x(torch.autograd.Function):
    """Bilinear x2, align_corners=False (reference generic_utils.py:96-105) with its adjoint as backward."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x):
        from . import ops
        with torch.no_grad():
            return ops.upsample2x(x.detach())

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        g = _dense_nhwc(g)
        b, c, h2, w2 = g.shape
        h, w = h2 // 2, w2 // 2
        out = empty_nhwc(b, c, h, w, g.device)
        if b > 0:
            gsb, gsp = _strides(g)
            osb, osp = _strides(out)
            with _lib.on_device(g.device):
                rc = _lib.lib().sr_upsample2x_bwd_nhwc(_lib.ptr(g), gsb, gsp, _lib.ptr(out), osb, osp, b, h, w, c,
                                                       _lib.stream_ptr(g.device))
            _lib.check(rc, "sr_upsample2x_bwd_nhwc")
        return out


class _Exp(torch.autograd.Function):
    """depth = exp(log_depth) (reference depth_model.py:392-400); backward = grad * depth."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x):
        from . import ops
        with torch.no_grad():
            y = ops.exp(x.detach())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous() if y.is_contiguous() else g.contiguous(memory_format=torch.channels_last)
        if g.stride() != y.stride():
            g = g.contiguous()
            y = y.contiguous()
        out = torch.empty_like(y)
        with _lib.on_device(y.device):
            rc = _lib.lib().sr_mul_fwd(_lib.ptr(g), _lib.ptr(y), _lib.ptr(out), y.numel(), _lib.stream_ptr(y.device))
        _lib.check(rc, "sr_mul_fwd")
        return out


def conv_bias_act(x, conv: nn.Conv2d, residual=None, slope=None):
    if conv.padding_mode != "zeros" or conv.groups != 1 or conv.dilation != (1, 1) or \
            tuple(conv.padding) != (conv.kernel_size[0] // 2,) * 2 or conv.kernel_size[0] not in (1, 3):
        raise _lib.HipLibraryError(f"unsupported Conv2d configuration for the HIP training path: {conv}")
    return _ConvBiasAct.apply(x, conv.weight, conv.bias, residual, conv.stride[0], slope)


def basic_block(block, x):
    """Differentiable BasicBlock.forward (reference layers.py:68-85), norm_layer = Identity."""
    if not isinstance(block.bn1, nn.Identity) or not isinstance(block.bn2, nn.Identity):
        raise _lib.HipLibraryError("the HIP BasicBlock implements norm_layer=nn.Identity only (what SimpleRecon uses)")
    slope = block.relu.negative_slope
    t = conv_bias_act(x, block.conv1, slope=slope)
    identity = x if block.downsample is None else conv_bias_act(x, block.downsample[0])
    return conv_bias_act(t, block.conv2, residual=identity, slope=slope)


def upsample2x(x):
    return _Upsample2x.apply(x)


def exp(x):
    return _Exp.apply(x)
