"""Differentiable operators of the two ENCODERS (training path, SURVEY.md §8f "next" #3): the reference trains
ResnetMatchingEncoder (modules/networks.py:149-205) and the timm EfficientNetV2-S pyramid (depth_model.py:110-116) end
to end, BatchNorm in training mode (train.py:126-145).

torch.autograd is the tape; every arithmetic step, forward and backward, is a HIP kernel (csrc/sr_train.hip for the
normalisations / pooling / depthwise / squeeze-excite pieces, the MFMA conv kernels and csrc/sr_conv_bwd.hip for dense
convolutions -- autograd_ops._ConvBiasAct with explicit pads).  In training the encoders run UNFUSED: conv -> BatchNorm
(+ activation) are separate operators because batch statistics need the raw conv output; inference keeps the folded,
fused kernels.  Gradients are pinned to the reference's own autograd (tests/golden/grad_matching_encoder_*.npz) and, for
the third-party EfficientNetV2-S, to the ATen restatement of its public definition (tests/effnet_torch.py)."""
import ctypes as C

import torch
from torch import nn

from . import _lib, autograd_ops, ops
from .ops import _strides, _workspace, as_nhwc, empty_nhwc

ACT_NONE, ACT_SILU, ACT_SIGMOID = -1.0, -2.0, -3.0
_amp_fwd, _amp_bwd = autograd_ops._amp_fwd, autograd_ops._amp_bwd   # fp32 kernels under torch.autocast (see autograd_ops)
_stash, _unstash = autograd_ops._stash, autograd_ops._unstash       # 16-bit storage of saved activations under autocast


def _dense(t, name="tensor"):
    return autograd_ops._dense_nhwc(t if t.dtype == torch.float32 else t.float())


def _add_flat_(a, b):
    """a += b for two dense fp32 tensors of the same size (numel % 4 == 0), on the HIP add kernel."""
    n = a.numel()
    with _lib.on_device(a.device):
        rc = _lib.lib().sr_add_nhwc_fwd(_lib.ptr(a), n, 4, _lib.ptr(b), n, 4, _lib.ptr(a), n, 4, 1, 1, n // 4, 4,
                                        _lib.stream_ptr(a.device))
    _lib.check(rc, "sr_add_nhwc_fwd")
    return a


# ------------------------------------------------------------------------------------ normalisation + activation --
class _NormAct(torch.autograd.Function):
    """y = act(gamma * (x - mean) / sqrt(var + eps) + beta).  per_image: InstanceNorm2d (statistics per image); else
    BatchNorm2d.  train_stats: mean / var are computed from x here (and returned for the running-statistics update);
    otherwise the given tensors are used as constants (BatchNorm in eval mode)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, gamma, beta, mean, var, eps, act, per_image, train_stats):
        lib = _lib.lib()
        x = as_nhwc(x, "normalisation input")
        b, c, h, w = x.shape
        dev = x.device
        g = b if per_image else 1
        out = empty_nhwc(b, c, h, w, dev)
        if train_stats:
            mean = torch.empty((g, c), dtype=torch.float32, device=dev)
            var = torch.empty((g, c), dtype=torch.float32, device=dev)
        else:
            mean, var = mean.detach().reshape(1, c).contiguous(), var.detach().reshape(1, c).contiguous()
        if b > 0:
            xsb, xsp = _strides(x)
            osb, osp = _strides(out)
            nws = lib.sr_norm_workspace_bytes(b, h * w, c, int(per_image))
            ws = _workspace(dev, "norm", nws)
            st = _lib.stream_ptr(dev)
            with _lib.on_device(dev):
                if train_stats:
                    _lib.check(lib.sr_norm_stats_nhwc(_lib.ptr(x), xsb, xsp, b, h * w, c, int(per_image), _lib.ptr(mean),
                                                      _lib.ptr(var), _lib.ptr(ws), nws, st), "sr_norm_stats_nhwc")
                gd = gamma.detach().contiguous() if gamma is not None else None
                bd = beta.detach().contiguous() if beta is not None else None
                _lib.check(lib.sr_norm_act_fwd_nhwc(_lib.ptr(x), xsb, xsp, _lib.ptr(mean), _lib.ptr(var), C.c_float(eps),
                                                    _lib.ptr(gd), _lib.ptr(bd), C.c_float(act), int(per_image),
                                                    _lib.ptr(out), osb, osp, b, h * w, c, st), "sr_norm_act_fwd_nhwc")
        _stash(ctx, x, gamma, beta, mean, var)
        ctx.cfg = (eps, act, per_image, train_stats)
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    @_amp_bwd
    def backward(ctx, g, _gm, _gv):
        x, gamma, beta, mean, var = _unstash(ctx)
        eps, act, per_image, train_stats = ctx.cfg
        lib = _lib.lib()
        b, c, h, w = x.shape
        dev = x.device
        g = _dense(g)
        dx = empty_nhwc(b, c, h, w, dev)
        need_g = gamma is not None and ctx.needs_input_grad[1]
        need_b = beta is not None and ctx.needs_input_grad[2]
        d_gamma = torch.empty((c,), dtype=torch.float32, device=dev) if need_g else None
        d_beta = torch.empty((c,), dtype=torch.float32, device=dev) if need_b else None
        if b == 0:
            return dx, (d_gamma.zero_() if need_g else None), (d_beta.zero_() if need_b else None), None, None, None, \
                None, None, None
        gsb, gsp = _strides(g)
        xsb, xsp = _strides(x)
        dsb, dsp = _strides(dx)
        nws = lib.sr_norm_workspace_bytes(b, h * w, c, int(per_image))
        ws = _workspace(dev, "norm", nws)
        gd = gamma.detach().contiguous() if gamma is not None else None
        bd = beta.detach().contiguous() if beta is not None else None
        with _lib.on_device(dev):
            rc = lib.sr_norm_act_bwd_nhwc(_lib.ptr(g), gsb, gsp, _lib.ptr(x), xsb, xsp, _lib.ptr(mean), _lib.ptr(var),
                                          C.c_float(eps), _lib.ptr(gd), _lib.ptr(bd), C.c_float(act), int(per_image),
                                          int(train_stats), _lib.ptr(dx), dsb, dsp, _lib.ptr(d_gamma), _lib.ptr(d_beta), b,
                                          h * w, c, _lib.ptr(ws), nws, _lib.stream_ptr(dev))
        _lib.check(rc, "sr_norm_act_bwd_nhwc")
        return dx, d_gamma, d_beta, None, None, None, None, None, None


def batch_norm_act(x, bn: nn.BatchNorm2d, act=ACT_NONE):
    """act(BatchNorm2d(x)) with the module's mode: training = batch statistics (+ running-statistics update with the
    module's momentum, unbiased variance, like ATen), eval = running statistics as constants."""
    if not isinstance(bn, nn.BatchNorm2d):
        raise _lib.HipLibraryError(f"expected nn.BatchNorm2d, got {type(bn).__name__}")
    use_batch = bn.training or bn.running_mean is None
    gamma, beta = (bn.weight, bn.bias) if bn.affine else (None, None)
    y, mean, var = _NormAct.apply(x, gamma, beta, None if use_batch else bn.running_mean,
                                  None if use_batch else bn.running_var, float(bn.eps), float(act), False, use_batch)
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            n = x.shape[0] * x.shape[2] * x.shape[3]
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1.0 - m).add_(mean[0], alpha=m)
            bn.running_var.mul_(1.0 - m).add_(var[0], alpha=m * n / max(n - 1, 1))
    return y


def instance_norm_act(x, eps=1e-5, leaky=None):
    """[LeakyReLU](InstanceNorm2d(x)), affine=False (reference networks.py:192-195, 200-201)."""
    return _NormAct.apply(x, None, None, None, None, float(eps), ACT_NONE if leaky is None else float(leaky), True, True)[0]


# ------------------------------------------------------------------------------------ pooling / padding / stem ----
class _MaxBlurPool(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x):
        x = as_nhwc(x, "maxblurpool input")
        with torch.no_grad():
            y = ops.maxblurpool(x.detach())
        _stash(ctx, x)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (x,) = _unstash(ctx)
        lib = _lib.lib()
        b, c, h, w = x.shape
        g = _dense(g)
        dx = empty_nhwc(b, c, h, w, x.device)
        if b == 0:
            return dx
        nws = lib.sr_maxblurpool_bwd_workspace_bytes(b, h, w, c)
        ws = _workspace(x.device, "maxblurpool_bwd", nws)
        with _lib.on_device(x.device):
            rc = lib.sr_maxblurpool_bwd_nhwc(_lib.ptr(g), *_strides(g), _lib.ptr(x), *_strides(x), _lib.ptr(dx),
                                             *_strides(dx), b, h, w, c, _lib.ptr(ws), nws, _lib.stream_ptr(x.device))
        _lib.check(rc, "sr_maxblurpool_bwd_nhwc")
        return dx


def maxblurpool(x):
    return _MaxBlurPool.apply(x)


class _ReplicatePad(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, pad):
        x = as_nhwc(x, "replicate-pad input")
        b, c, h, w = x.shape
        y = empty_nhwc(b, c, h + 2 * pad, w + 2 * pad, x.device)
        if b > 0:
            with _lib.on_device(x.device):
                rc = _lib.lib().sr_replicate_pad_nhwc_fwd(_lib.ptr(x), *_strides(x), _lib.ptr(y), b, h, w, c, pad,
                                                          _lib.stream_ptr(x.device))
            _lib.check(rc, "sr_replicate_pad_nhwc_fwd")
        ctx.shape, ctx.pad = (b, c, h, w), pad
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        b, c, h, w = ctx.shape
        g = _dense(g)
        dx = empty_nhwc(b, c, h, w, g.device)
        if b > 0:
            with _lib.on_device(g.device):
                rc = _lib.lib().sr_replicate_pad_nhwc_bwd(_lib.ptr(g), _lib.ptr(dx), b, h, w, c, ctx.pad,
                                                          _lib.stream_ptr(g.device))
            _lib.check(rc, "sr_replicate_pad_nhwc_bwd")
        return dx, None


def replicate_pad(x, pad=1):
    return _ReplicatePad.apply(x, pad)


class _Stem7x7(torch.autograd.Function):
    """conv 7x7 / stride 2 / pad 3, 3 -> 64 (ResNet conv1, reference networks.py:176).  The image needs no gradient; the
    weight gradient is a 1x1-conv weight gradient over the unfolded input (sr_im2col7x7s2_nhwc + sr_conv_wgrad_nhwc)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, image, weight, conv):
        with torch.no_grad():
            y = ops.stem7x7(image.detach(), conv, bn=None, leaky=None)
        ctx.save_for_backward(image)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (image,) = ctx.saved_tensors
        lib = _lib.lib()
        dev = image.device
        b, _, h, w = image.shape
        g = _dense(g)
        ho, wo = g.shape[2], g.shape[3]
        kp = 160   # 147 taps padded to a multiple of 32
        if not ctx.needs_input_grad[1]:
            return None, None, None
        if b == 0:
            return None, torch.zeros((64, 3, 7, 7), dtype=torch.float32, device=dev), None
        st = _lib.stream_ptr(dev)
        # unfold a few images at a time into ONE reused buffer of <= 256 MB: [n, Ho, Wo, 160] floats (49 MB per 640x480
        # image); the weight-gradient partials of the chunks are added in chunk order
        per = max(1, min(b, (1 << 28) // max(ho * wo * kp * 4, 1)))
        acc = None
        col_buf = torch.empty((per, ho, wo, kp), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            for i0 in range(0, b, per):
                n = min(per, b - i0)
                col = col_buf[:n]
                img = image[i0:i0 + n]
                sb, sc, sy, sx = img.stride()
                _lib.check(lib.sr_im2col7x7s2_nhwc(_lib.ptr(img), sb, sc, sy, sx, _lib.ptr(col), n, h, w, kp, st),
                           "sr_im2col7x7s2_nhwc")
                gi = g[i0:i0 + n]
                gsb, gsp = _strides(gi)
                dw = torch.empty((64, kp), dtype=torch.float32, device=dev)
                nws = lib.sr_conv_wgrad_workspace_bytes(n, ho, wo, kp, 64, 1, 1)
                ws = _workspace(dev, "wgrad", nws)
                _lib.check(lib.sr_conv_wgrad_nhwc(_lib.ptr(col), ho * wo * kp, kp, _lib.ptr(gi), gsb, gsp, _lib.ptr(dw), n, ho,
                                                  wo, kp, 64, 1, 1, _lib.ptr(ws), nws, st), "sr_conv_wgrad_nhwc")
                acc = dw if acc is None else _add_flat_(acc, dw)
        d_w = acc[:, :147].reshape(64, 3, 7, 7).contiguous()
        return None, d_w, None


def stem7x7(image, conv: nn.Conv2d):
    if conv.bias is not None:
        raise _lib.HipLibraryError("the HIP stem trains the bias-free ResNet conv1 only")
    if torch.is_grad_enabled() and image.requires_grad:
        # like geometry._data: a gradient this path does not compute must not come back silently as None
        raise _lib.HipLibraryError("the HIP matching-encoder stem has no gradient w.r.t. the image (the reference trains "
                                   "on images as data); detach() the image or use the torch module")
    return _Stem7x7.apply(image, conv.weight, conv)


# ------------------------------------------------------------------------------------ dense convolution -----------
def conv(x, conv_mod: nn.Conv2d, pads=None, residual=None, slope=None):
    """conv_mod(x) (+ residual, LeakyReLU(slope)) with optional explicit (top, left, bottom, right) zero padding."""
    k = conv_mod.kernel_size[0]
    # (a module with another padding mode is fine when the caller pads itself and passes explicit `pads`)
    if (conv_mod.padding_mode != "zeros" and pads is None) or conv_mod.groups != 1 or conv_mod.dilation != (1, 1) \
            or k not in (1, 3) or conv_mod.kernel_size[0] != conv_mod.kernel_size[1] \
            or conv_mod.stride[0] != conv_mod.stride[1]:
        raise _lib.HipLibraryError(f"unsupported Conv2d configuration for the HIP training path: {conv_mod}")
    if pads is None:
        pads = (conv_mod.padding[0], conv_mod.padding[1], conv_mod.padding[0], conv_mod.padding[1])
    pads = tuple(int(p) for p in pads)
    if pads == (k // 2,) * 4:
        pads = None
    return autograd_ops._ConvBiasAct.apply(x, conv_mod.weight, conv_mod.bias, residual, conv_mod.stride[0], slope, pads)


# ------------------------------------------------------------------------------------ depthwise 3x3 ---------------
class _DwConv3x3(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, conv, pads):
        x = as_nhwc(x, "depthwise conv input")
        with torch.no_grad():
            y = ops.dwconv3x3(x.detach(), conv, bn=None, tf_same=False, pads=pads)
        _stash(ctx, x, weight)
        ctx.cfg = (conv.stride[0], pads)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x, weight = _unstash(ctx)
        s, pads = ctx.cfg
        lib = _lib.lib()
        b, c, h, w = x.shape
        g = _dense(g)
        ho, wo = g.shape[2], g.shape[3]
        dev = x.device
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = empty_nhwc(b, c, h, w, dev) if need_x else None
        dw = torch.empty((c, 1, 3, 3), dtype=torch.float32, device=dev) if need_w else None
        if b == 0:
            return dx, (dw.zero_() if need_w else None), None, None
        nws = lib.sr_dwconv3x3_bwd_workspace_bytes(b, ho, wo, c)
        ws = _workspace(dev, "dw_bwd", nws)
        wd = weight.detach().reshape(c, 9).contiguous()
        with _lib.on_device(dev):
            rc = lib.sr_dwconv3x3_bwd_nhwc(_lib.ptr(g), *_strides(g), _lib.ptr(x), *_strides(x), _lib.ptr(wd), _lib.ptr(dx),
                                           _lib.ptr(dw), b, h, w, c, s, pads[0], pads[1], ho, wo, _lib.ptr(ws), nws,
                                           _lib.stream_ptr(dev))
        _lib.check(rc, "sr_dwconv3x3_bwd_nhwc")
        return dx, dw, None, None


def dwconv3x3(x, conv_mod: nn.Conv2d, pads):
    if conv_mod.bias is not None or conv_mod.kernel_size != (3, 3) or conv_mod.groups != conv_mod.in_channels:
        raise _lib.HipLibraryError(f"unsupported depthwise Conv2d for the HIP training path: {conv_mod}")
    return _DwConv3x3.apply(x, conv_mod.weight, conv_mod, tuple(int(p) for p in pads))


# ------------------------------------------------------------------------------------ squeeze-excite ---------------
class _SqueezeExcite(torch.autograd.Function):
    """y = x * sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2) (timm SqueezeExcite: 1x1 convs on the pooled vector)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w1, b1, w2, b2):
        lib = _lib.lib()
        x = as_nhwc(x, "squeeze-excite input")
        b, c, h, w = x.shape
        dev = x.device
        rd = w1.shape[0]
        w1d, w2d = w1.detach().reshape(rd, c).contiguous(), w2.detach().reshape(c, rd).contiguous()
        pooled = torch.empty((b, c), dtype=torch.float32, device=dev)
        pre1, hid = (torch.empty((b, rd), dtype=torch.float32, device=dev) for _ in range(2))
        pre2, gate = (torch.empty((b, c), dtype=torch.float32, device=dev) for _ in range(2))
        y = empty_nhwc(b, c, h, w, dev)
        if b > 0:
            nws = lib.sr_norm_workspace_bytes(b, h * w, c, 1)
            ws = _workspace(dev, "norm", nws)
            st = _lib.stream_ptr(dev)
            with _lib.on_device(dev):
                _lib.check(lib.sr_rowsum_nhwc(_lib.ptr(x), *_strides(x), None, 0, 0, b, h * w, c, C.c_float(1.0 / (h * w)),
                                              _lib.ptr(pooled), _lib.ptr(ws), nws, st), "sr_rowsum_nhwc")
                _lib.check(lib.sr_small_linear_fwd(_lib.ptr(pooled), _lib.ptr(w1d), _lib.ptr(b1.detach().contiguous()),
                                                   _lib.ptr(pre1), _lib.ptr(hid), b, c, rd, C.c_float(ACT_SILU), st),
                           "sr_small_linear_fwd")
                _lib.check(lib.sr_small_linear_fwd(_lib.ptr(hid), _lib.ptr(w2d), _lib.ptr(b2.detach().contiguous()),
                                                   _lib.ptr(pre2), _lib.ptr(gate), b, rd, c, C.c_float(ACT_SIGMOID), st),
                           "sr_small_linear_fwd")
                _lib.check(lib.sr_scale_channels_nhwc_fwd(_lib.ptr(x), *_strides(x), _lib.ptr(gate), _lib.ptr(y), *_strides(y),
                                                          b, h, w, c, st), "sr_scale_channels_nhwc_fwd")
        _stash(ctx, x, w1d, w2d, pooled, pre1, hid, pre2, gate)
        ctx.shapes = (tuple(w1.shape), tuple(w2.shape))
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x, w1d, w2d, pooled, pre1, hid, pre2, gate = _unstash(ctx)
        lib = _lib.lib()
        b, c, h, w = x.shape
        rd = w1d.shape[0]
        dev = x.device
        g = _dense(g)
        dx = empty_nhwc(b, c, h, w, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        dgate, dpool = torch.empty((b, c), **f32), torch.empty((b, c), **f32)
        dhid = torch.empty((b, rd), **f32)
        dw1, db1, dw2, db2 = torch.empty((rd, c), **f32), torch.empty((rd,), **f32), torch.empty((c, rd), **f32), \
            torch.empty((c,), **f32)
        if b == 0:
            return dx, dw1.zero_().reshape(ctx.shapes[0]), db1.zero_(), dw2.zero_().reshape(ctx.shapes[1]), db2.zero_()
        nws = lib.sr_norm_workspace_bytes(b, h * w, c, 1)
        ws = _workspace(dev, "norm", nws)
        st = _lib.stream_ptr(dev)
        with _lib.on_device(dev):
            _lib.check(lib.sr_rowsum_nhwc(_lib.ptr(x), *_strides(x), _lib.ptr(g), *_strides(g), b, h * w, c, C.c_float(1.0),
                                          _lib.ptr(dgate), _lib.ptr(ws), nws, st), "sr_rowsum_nhwc")
            _lib.check(lib.sr_small_linear_bwd(_lib.ptr(dgate), _lib.ptr(pre2), _lib.ptr(hid), _lib.ptr(w2d), _lib.ptr(dhid),
                                               _lib.ptr(dw2), _lib.ptr(db2), b, rd, c, C.c_float(ACT_SIGMOID), st),
                       "sr_small_linear_bwd")
            _lib.check(lib.sr_small_linear_bwd(_lib.ptr(dhid), _lib.ptr(pre1), _lib.ptr(pooled), _lib.ptr(w1d), _lib.ptr(dpool),
                                               _lib.ptr(dw1), _lib.ptr(db1), b, c, rd, C.c_float(ACT_SILU), st),
                       "sr_small_linear_bwd")
            _lib.check(lib.sr_scale_bwd_nhwc(_lib.ptr(g), *_strides(g), _lib.ptr(gate), _lib.ptr(dpool),
                                             C.c_float(1.0 / (h * w)), _lib.ptr(dx), b, h * w, c, st), "sr_scale_bwd_nhwc")
        return dx, dw1.reshape(ctx.shapes[0]), db1, dw2.reshape(ctx.shapes[1]), db2


def squeeze_excite(x, conv_reduce: nn.Conv2d, conv_expand: nn.Conv2d):
    """x * sigmoid(conv_expand(silu(conv_reduce(mean_hw(x))))) (timm SqueezeExcite with SiLU / sigmoid)."""
    if conv_reduce.bias is None or conv_expand.bias is None or conv_reduce.kernel_size != (1, 1) \
            or conv_expand.kernel_size != (1, 1):
        raise _lib.HipLibraryError("squeeze-excite expects biased 1x1 convs C -> rd -> C")
    return _SqueezeExcite.apply(x, conv_reduce.weight, conv_reduce.bias, conv_expand.weight, conv_expand.bias)


class _AddAct(torch.autograd.Function):
    """act(a + b): the residual join of a block (identity skip), activation optional."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, a, b, act):
        a, b = _dense(as_nhwc(a, "a")), _dense(as_nhwc(b, "b"))
        if a.shape != b.shape:
            raise ValueError(f"add: shapes differ, {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty_like(a)
        pre = torch.empty_like(a) if act != ACT_NONE else None
        if a.numel() > 0:
            with _lib.on_device(a.device):
                rc = _lib.lib().sr_add_act_fwd(_lib.ptr(a), _lib.ptr(b), _lib.ptr(pre), _lib.ptr(out), a.numel(), C.c_float(act),
                                               _lib.stream_ptr(a.device))
            _lib.check(rc, "sr_add_act_fwd")
        ctx.act = act
        if pre is not None:
            _stash(ctx, pre)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        if ctx.act == ACT_NONE:
            return g, g, None
        (pre,) = _unstash(ctx)
        g = _dense(g)
        gz = torch.empty_like(pre)
        if pre.numel() > 0:
            with _lib.on_device(pre.device):
                rc = _lib.lib().sr_act_in_bwd(_lib.ptr(g), _lib.ptr(pre), _lib.ptr(gz), pre.numel(), C.c_float(ctx.act),
                                              _lib.stream_ptr(pre.device))
            _lib.check(rc, "sr_act_in_bwd")
        return gz, gz, None


def add(a, b, act=ACT_NONE):
    return _AddAct.apply(a, b, float(act))
