"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps oracle/liboracle.so (plain C, see sr_oracle_body.h) and
restates, on top of its conv primitives, the reference's module compositions:

  BasicBlock.forward      modules/layers.py:68-85
  CVEncoder.forward       modules/networks.py:120-127
  DepthDecoderPP.forward  modules/networks.py:75-96

Weights are passed as a flat {state_dict key: numpy array} mapping that uses the
reference's own parameter names (e.g. "convs.ds_conv_0.conv1.weight").
Parity: pinned against tests/golden/ (outputs of the reference modules run on CPU).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("sr_oracle.c", "sr_oracle_body.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
    return _LIB


def num_threads():
    return int(lib().sr_oracle_num_threads())


def set_threads(n):
    lib().sr_oracle_set_threads(int(n))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _planes_arg(planes, B, D, h, w):
    """planes: [B,D] or [B,D,h,w] (numpy, fp32) -> (array, strides in elements)."""
    p = _f32(planes)
    if p.ndim == 2:
        assert p.shape == (B, D)
        return p, (D, 1, 0, 0)
    assert p.shape == (B, D, h, w)
    return p, (D * h * w, h * w, w, 1)


def _dt(precision):
    return (np.float32, "_f32") if precision == "f32" else (np.float64, "_f64")


def dot_volume(cur, src, K_src, T_src_cur, invK_cur, planes, want_mask=False, precision="f32"):
    cur, src = _f32(cur), _f32(src)
    B, K, Cc, h, w = src.shape
    K_src, T_src_cur, invK_cur = _f32(K_src), _f32(T_src_cur), _f32(invK_cur)
    D = planes.shape[1]
    p, ps = _planes_arg(planes, B, D, h, w)
    dt, sfx = _dt(precision)
    cv = np.empty((B, D, h, w), dt)
    low = np.empty((B, h, w), dt)
    mask = np.empty((B, h, w), np.uint8) if want_mask else None
    fn = getattr(lib(), "sr_oracle_dot_volume" + sfx)
    rc = fn(_ptr(cur), _ptr(src), _ptr(K_src), _ptr(T_src_cur), _ptr(invK_cur), _ptr(p),
            C.c_long(ps[0]), C.c_long(ps[1]), C.c_long(ps[2]), C.c_long(ps[3]),
            B, K, Cc, h, w, D, _ptr(cv), _ptr(low), _ptr(mask))
    assert rc == 0, rc
    return cv, low, (mask.astype(bool) if want_mask else None)


def dot_volume_backward(grad_cv, cur, src, K_src, T_src_cur, invK_cur, planes, precision="f32"):
    """(d_cur_feats, d_src_feats) of L for grad_cv = dL/d cost_volume [B,D,h,w] (sr_oracle_dot_volume_bwd)."""
    cur, src, g = _f32(cur), _f32(src), _f32(grad_cv)
    B, K, Cc, h, w = src.shape
    K_src, T_src_cur, invK_cur = _f32(K_src), _f32(T_src_cur), _f32(invK_cur)
    D = planes.shape[1]
    p, ps = _planes_arg(planes, B, D, h, w)
    dt, sfx = _dt(precision)
    d_cur, d_src = np.empty((B, Cc, h, w), dt), np.empty((B, K, Cc, h, w), dt)
    rc = getattr(lib(), "sr_oracle_dot_volume_bwd" + sfx)(
        _ptr(g), _ptr(cur), _ptr(src), _ptr(K_src), _ptr(T_src_cur), _ptr(invK_cur), _ptr(p), C.c_long(ps[0]),
        C.c_long(ps[1]), C.c_long(ps[2]), C.c_long(ps[3]), B, K, Cc, h, w, D, _ptr(d_cur), _ptr(d_src))
    assert rc == 0, rc
    return d_cur, d_src


def pose_features(T_cur_src):
    """pose_distance (utils/geometry_utils.py:178-191) -> [B,K,3] = (dist, R_measure, t_measure), fp32."""
    T = _f32(T_cur_src)
    R = T[..., :3, :3]
    t = T[..., :3, 3]
    tr = (R[..., 0, 0] + R[..., 1, 1]) + R[..., 2, 2]
    r_m = np.sqrt(np.float32(2) * (np.float32(1) - np.minimum(np.float32(3), tr) / np.float32(3)))
    t_m = np.sqrt((t * t).sum(-1, dtype=np.float32))
    dist = np.sqrt(t_m ** 2 + r_m ** 2)
    return np.stack([dist, r_m, t_m], -1).astype(np.float32)


def mlp_volume(cur, src, K_src, T_src_cur, T_cur_src, invK_cur, planes, mlp, want_mask=False,
               precision="f32", pose_feats=None):
    """mlp: dict with W1,b1,W2,b2,W3,b3 (numpy)."""
    cur, src = _f32(cur), _f32(src)
    B, K, Cc, h, w = src.shape
    K_src, T_src_cur, T_cur_src, invK_cur = _f32(K_src), _f32(T_src_cur), _f32(T_cur_src), _f32(invK_cur)
    D = planes.shape[1]
    p, ps = _planes_arg(planes, B, D, h, w)
    pf = _f32(pose_feats if pose_feats is not None else pose_features(T_cur_src))
    W1, b1, W2, b2, W3, b3 = (_f32(mlp[k]) for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
    Hd = W1.shape[0]
    assert W1.shape[1] == Cc * (K + 1) + 10 * K + 4, (W1.shape, Cc, K)
    dt, sfx = _dt(precision)
    cv = np.empty((B, D, h, w), dt)
    low = np.empty((B, h, w), dt)
    mask = np.empty((B, h, w), np.uint8) if want_mask else None
    fn = getattr(lib(), "sr_oracle_mlp_volume" + sfx)
    rc = fn(_ptr(cur), _ptr(src), _ptr(K_src), _ptr(T_src_cur), _ptr(T_cur_src), _ptr(invK_cur), _ptr(pf),
            _ptr(p), C.c_long(ps[0]), C.c_long(ps[1]), C.c_long(ps[2]), C.c_long(ps[3]),
            _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(W3), _ptr(b3),
            B, K, Cc, h, w, D, Hd, _ptr(cv), _ptr(low), _ptr(mask))
    assert rc == 0, rc
    return cv, low, (mask.astype(bool) if want_mask else None)


def mlp_volume_backward(grad_cv, cur, src, K_src, T_src_cur, T_cur_src, invK_cur, planes, mlp, precision="f32"):
    """Gradients of L w.r.t. (cur_feats, src_feats, W1, b1, W2, b2, W3, b3) for grad_cv = dL/d cost_volume of the
    metadata-MLP volume (sr_oracle_mlp_volume_bwd); returns a dict keyed d_cur_feats, d_src_feats, dW1, ..."""
    cur, src, g = _f32(cur), _f32(src), _f32(grad_cv)
    B, K, Cc, h, w = src.shape
    K_src, T_src_cur, T_cur_src, invK_cur = _f32(K_src), _f32(T_src_cur), _f32(T_cur_src), _f32(invK_cur)
    D = planes.shape[1]
    p, ps = _planes_arg(planes, B, D, h, w)
    pf = _f32(pose_features(T_cur_src))
    W1, b1, W2, b2, W3, b3 = (_f32(mlp[k]) for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
    Hd, Cin = W1.shape
    assert Cin == Cc * (K + 1) + 10 * K + 4, (W1.shape, Cc, K)
    dt, sfx = _dt(precision)
    out = dict(d_cur_feats=np.empty((B, Cc, h, w), dt), d_src_feats=np.empty((B, K, Cc, h, w), dt),
               dW1=np.empty((Hd, Cin), dt), db1=np.empty((Hd,), dt), dW2=np.empty((Hd, Hd), dt),
               db2=np.empty((Hd,), dt), dW3=np.empty((1, Hd), dt), db3=np.empty((1,), dt))
    rc = getattr(lib(), "sr_oracle_mlp_volume_bwd" + sfx)(
        _ptr(g), _ptr(cur), _ptr(src), _ptr(K_src), _ptr(T_src_cur), _ptr(T_cur_src), _ptr(invK_cur), _ptr(pf),
        _ptr(p), C.c_long(ps[0]), C.c_long(ps[1]), C.c_long(ps[2]), C.c_long(ps[3]),
        _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(W3), _ptr(b3), B, K, Cc, h, w, D, Hd,
        *[_ptr(out[k]) for k in ("d_cur_feats", "d_src_feats", "dW1", "db1", "dW2", "db2", "dW3", "db3")])
    assert rc == 0, rc
    return out


def mlp_input(cur, src, K_src, T_src_cur, T_cur_src, invK_cur, d, b, y, x, precision="f32"):
    cur, src = _f32(cur), _f32(src)
    B, K, Cc, h, w = src.shape
    pf = pose_features(T_cur_src)
    dt, sfx = _dt(precision)
    out = np.empty((Cc * (K + 1) + 10 * K + 4,), dt)
    fn = getattr(lib(), "sr_oracle_mlp_input" + sfx)
    dd = C.c_float(d) if precision == "f32" else C.c_double(d)
    rc = fn(_ptr(cur), _ptr(src), _ptr(_f32(K_src)), _ptr(_f32(T_src_cur)), _ptr(_f32(T_cur_src)),
            _ptr(_f32(invK_cur)), _ptr(pf), dd, b, y, x, K, Cc, h, w, _ptr(out))
    assert rc == 0
    return out


# ---------------------------------------------------------------- conv stack --

def conv2d(x, wgt, bias=None, stride=1, pad=None, residual=None, leaky=None, precision="f32"):
    dt, sfx = _dt(precision)
    x = np.ascontiguousarray(x, dtype=dt)
    wgt = _f32(wgt)
    B, Ci, H, W = x.shape
    Co, Ci2, k, _ = wgt.shape
    assert Ci == Ci2, (x.shape, wgt.shape)
    if pad is None:
        pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = np.empty((B, Co, Ho, Wo), dt)
    b = _f32(bias) if bias is not None else None
    r = np.ascontiguousarray(residual, dtype=dt) if residual is not None else None
    slope = -1.0 if leaky is None else float(leaky)
    fn = getattr(lib(), "sr_oracle_conv2d" + sfx)
    sl = C.c_float(slope) if precision == "f32" else C.c_double(slope)
    rc = fn(_ptr(x), _ptr(wgt), _ptr(b), _ptr(r), B, Ci, H, W, Co, k, stride, pad, sl, _ptr(out))
    assert rc == 0, rc
    return out


def upsample2x(x, precision="f32"):
    dt, sfx = _dt(precision)
    x = np.ascontiguousarray(x, dtype=dt)
    B, Cc, H, W = x.shape
    out = np.empty((B, Cc, 2 * H, 2 * W), dt)
    rc = getattr(lib(), "sr_oracle_upsample2x" + sfx)(_ptr(x), B, Cc, H, W, _ptr(out))
    assert rc == 0
    return out


def basic_block(x, sd, prefix, stride=1, precision="f32"):
    """BasicBlock.forward (modules/layers.py:68-85) with norm_layer=Identity (bias=True),
    LeakyReLU(0.2); downsample = conv1x1 (stride 1) or conv3x3 (stride 2) (layers.py:58-65)."""
    out = conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], stride=stride, leaky=0.2,
                 precision=precision)
    ident = x
    if (prefix + "downsample.0.weight") in sd:
        ident = conv2d(x, sd[prefix + "downsample.0.weight"], sd[prefix + "downsample.0.bias"], stride=stride,
                       precision=precision)
    return conv2d(out, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], residual=ident, leaky=0.2,
                  precision=precision)


# Backward of the conv stack -- groundwork for SURVEY.md §8f "next" #3 beyond the cost volumes: what autograd computes
# through nn.Conv2d (+ bias), LeakyReLU and the residual add of BasicBlock (modules/layers.py:68-85).  numpy, fp64
# accumulation inside einsum; pinned to the reference's autograd in tests/golden/grad_block_*.npz.

def conv2d_backward(x, wgt, gy, stride=1, pad=None):
    """(dx, dw, db) of y = conv2d(x, wgt) + bias for upstream gradient gy (zero padding, square kernel)."""
    x, wgt, gy = (np.asarray(a, dtype=np.float64) for a in (x, wgt, gy))
    Co, Ci, k, _ = wgt.shape
    if pad is None:
        pad = k // 2
    B, _, H, W = x.shape
    Ho, Wo = gy.shape[2:]
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(wgt)
    for ky in range(k):
        for kx in range(k):
            ys, xs = slice(ky, ky + stride * Ho, stride), slice(kx, kx + stride * Wo, stride)
            dw[:, :, ky, kx] = np.einsum("bohw,bihw->oi", gy, xp[:, :, ys, xs])
            dxp[:, :, ys, xs] += np.einsum("bohw,oi->bihw", gy, wgt[:, :, ky, kx])
    dx = dxp[:, :, pad:pad + H, pad:pad + W]
    return dx, dw, gy.sum(axis=(0, 2, 3))


def basic_block_backward(x, sd, prefix, gy, stride=1):
    """Gradients of BasicBlock.forward (norm_layer=Identity, LeakyReLU(0.2)) w.r.t. its input and parameters:
    returns {"x": dx, "conv1.weight": ..., "conv1.bias": ..., "conv2.*", ["downsample.0.*"]} (float64)."""
    slope = 0.2
    z1 = conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], stride=stride, precision="f64")
    a1 = np.where(z1 > 0, z1, z1 * slope)
    has_ds = (prefix + "downsample.0.weight") in sd
    ident = conv2d(x, sd[prefix + "downsample.0.weight"], sd[prefix + "downsample.0.bias"], stride=stride,
                   precision="f64") if has_ds else np.asarray(x, dtype=np.float64)
    z2 = conv2d(a1, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], precision="f64") + ident
    g2 = np.asarray(gy, dtype=np.float64) * np.where(z2 > 0, 1.0, slope)          # through the final LeakyReLU
    out = {}
    da1, out["conv2.weight"], out["conv2.bias"] = conv2d_backward(a1, sd[prefix + "conv2.weight"], g2)
    g1 = da1 * np.where(z1 > 0, 1.0, slope)
    dx, out["conv1.weight"], out["conv1.bias"] = conv2d_backward(x, sd[prefix + "conv1.weight"], g1, stride=stride)
    if has_ds:
        dxi, out["downsample.0.weight"], out["downsample.0.bias"] = conv2d_backward(
            x, sd[prefix + "downsample.0.weight"], g2, stride=stride)
        dx = dx + dxi
    else:
        dx = dx + g2
    out["x"] = dx
    return out


def cv_encoder(x, img_feats, sd, precision="f32"):
    """CVEncoder.forward (modules/networks.py:120-127)."""
    outs = []
    for i in range(len(img_feats)):
        x = basic_block(x, sd, f"convs.ds_conv_{i}.", stride=1 if i == 0 else 2, precision=precision)
        x = np.concatenate([x, np.asarray(img_feats[i], dtype=x.dtype)], axis=1)
        x = basic_block(x, sd, f"convs.conv_{i}.0.", precision=precision)
        x = basic_block(x, sd, f"convs.conv_{i}.1.", precision=precision)
        outs.append(x)
    return outs


def cv_encoder_backward(x, img_feats, sd, gouts):
    """Backward of CVEncoder.forward for upstream gradients gouts[i] = dL/d outs[i]: returns (dx, [d img_feats[i]],
    {parameter name: gradient}) in float64.  Level i's output feeds level i+1, so its gradient is gouts[i] plus what
    flows back from the next level."""
    acts = []                      # (ds input, ds output = concat part, concat, conv_i.0 output)
    h = np.asarray(x, dtype=np.float64)
    for i in range(len(img_feats)):
        a = basic_block(h, sd, f"convs.ds_conv_{i}.", stride=1 if i == 0 else 2, precision="f64")
        c = np.concatenate([a, np.asarray(img_feats[i], dtype=np.float64)], axis=1)
        m = basic_block(c, sd, f"convs.conv_{i}.0.", precision="f64")
        o = basic_block(m, sd, f"convs.conv_{i}.1.", precision="f64")
        acts.append((h, a, c, m))
        h = o
    grads, dfeats = {}, [None] * len(img_feats)
    g = np.zeros_like(h)
    for i in reversed(range(len(img_feats))):
        hin, a, c, m = acts[i]
        g = g + np.asarray(gouts[i], dtype=np.float64)
        for prefix, inp, stride in ((f"convs.conv_{i}.1.", m, 1), (f"convs.conv_{i}.0.", c, 1)):
            r = basic_block_backward(inp, sd, prefix, g, stride=stride)
            g = r.pop("x")
            grads.update({prefix + k: v for k, v in r.items()})
        dfeats[i] = g[:, a.shape[1]:]
        r = basic_block_backward(hin, sd, f"convs.ds_conv_{i}.", g[:, :a.shape[1]], stride=1 if i == 0 else 2)
        g = r.pop("x")
        grads.update({f"convs.ds_conv_{i}." + k: v for k, v in r.items()})
    return g, dfeats, grads


def depth_decoder_pp(feats, sd, precision="f32"):
    """DepthDecoderPP.forward (modules/networks.py:75-96); double_basic_block naming
    (networks.py:13-17): Sequential(0: BasicBlock, conv_0: BasicBlock)."""
    dt, _ = _dt(precision)
    prev = [np.asarray(f, dtype=dt) for f in feats]
    outputs, depth_outputs = [], {}
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            inputs = [basic_block(prev[i], sd, f"convs.right_conv_{i}{j-1}.", precision=precision)]
            inputs.append(upsample2x(basic_block(prev[i + 1], sd, f"convs.diag_conv_{i+1}{j-1}.",
                                                 precision=precision), precision))
            if i + j != 4:
                inputs.append(upsample2x(basic_block(outputs[-1], sd, f"convs.up_conv_{i+1}{j}.",
                                                     precision=precision), precision))
            o = np.concatenate(inputs, axis=1)
            o = basic_block(o, sd, f"convs.in_conv_{i}{j}.0.", precision=precision)
            o = basic_block(o, sd, f"convs.in_conv_{i}{j}.conv_0.", precision=precision)
            outputs.append(o)
            hd = o
            if i != 0:
                hd = basic_block(o, sd, f"convs.output_{i}.0.", precision=precision)
            depth_outputs[f"log_depth_pred_s{i}_b1hw"] = conv2d(
                hd, sd[f"convs.output_{i}.1.weight"], sd[f"convs.output_{i}.1.bias"], precision=precision)
        prev = outputs[::-1]
    return depth_outputs


# Backward of DepthDecoderPP: the UNet++ graph re-uses nodes along several paths, so instead of a hand-reversed loop
# the forward is replayed on a tiny tape (float64) whose entries know their own backward (basic_block_backward /
# conv2d_backward above, the transposes of bilinear x2 and of the channel concat); pinned to the reference's autograd.

class _Node:
    __slots__ = ("v", "g", "back")

    def __init__(self, v, back=None):
        self.v, self.g, self.back = np.asarray(v, dtype=np.float64), None, back

    def accum(self, g):
        self.g = g if self.g is None else self.g + g


def _up2_axis_weights(n):
    """Taps of nn.Upsample(scale 2, bilinear, align_corners=False) along one axis: out[o] = w0 in[i0] + w1 in[i1]."""
    o = np.arange(2 * n)
    src = (o + 0.5) / 2.0 - 0.5
    src = np.maximum(src, 0.0)                    # ATen clamps negative source coordinates to 0
    i0 = np.minimum(np.floor(src).astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    w1 = src - i0
    return i0, i1, 1.0 - w1, w1


def _up2_axis(x, axis, transpose=False):
    n = x.shape[axis] // 2 if transpose else x.shape[axis]
    i0, i1, w0, w1 = _up2_axis_weights(n)
    shp = [1] * x.ndim
    shp[axis] = -1
    if not transpose:
        return np.take(x, i0, axis=axis) * w0.reshape(shp) + np.take(x, i1, axis=axis) * w1.reshape(shp)
    out = np.zeros(x.shape[:axis] + (n,) + x.shape[axis + 1:], dtype=x.dtype)
    xm = np.moveaxis(x, axis, 0)
    om = np.moveaxis(out, axis, 0)
    np.add.at(om, i0, xm * w0.reshape((-1,) + (1,) * (x.ndim - 1)))
    np.add.at(om, i1, xm * w1.reshape((-1,) + (1,) * (x.ndim - 1)))
    return out


def upsample2x_backward(gy):
    """Transpose of upsample2x (bilinear x2, align_corners=False) applied to gy [B,C,2H,2W] -> [B,C,H,W]."""
    return _up2_axis(_up2_axis(np.asarray(gy, dtype=np.float64), 3, True), 2, True)


def depth_decoder_pp_backward(feats, sd, gouts):
    """Backward of DepthDecoderPP.forward for gouts = {"log_depth_pred_s{i}_b1hw": dL/d output}: returns
    ([d feats[i]], {parameter name: gradient}) in float64."""
    tape, grads = [], {}

    def add_grad(name, g):
        grads[name] = grads[name] + g if name in grads else g

    def leaf(v):
        return _Node(v)

    def bb(x, prefix, stride=1):
        node = _Node(basic_block(x.v, sd, prefix, stride=stride, precision="f64"))

        def back():
            r = basic_block_backward(x.v, sd, prefix, node.g, stride=stride)
            x.accum(r.pop("x"))
            for k, v in r.items():
                add_grad(prefix + k, v)
        node.back = back
        tape.append(node)
        return node

    def up(x):
        node = _Node(_up2_axis(_up2_axis(x.v, 2), 3))
        node.back = lambda: x.accum(upsample2x_backward(node.g))
        tape.append(node)
        return node

    def cat(xs):
        node = _Node(np.concatenate([x.v for x in xs], axis=1))

        def back():
            c0 = 0
            for x in xs:
                x.accum(node.g[:, c0:c0 + x.v.shape[1]])
                c0 += x.v.shape[1]
        node.back = back
        tape.append(node)
        return node

    def head(x, name):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        node = _Node(conv2d(x.v, w, b, precision="f64"))

        def back():
            dx, dw, db = conv2d_backward(x.v, w, node.g)
            x.accum(dx)
            add_grad(name + ".weight", dw)
            add_grad(name + ".bias", db)
        node.back = back
        tape.append(node)
        return node

    inputs = [leaf(f) for f in feats]
    prev, outputs, final = list(inputs), [], {}
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            parts = [bb(prev[i], f"convs.right_conv_{i}{j-1}."), up(bb(prev[i + 1], f"convs.diag_conv_{i+1}{j-1}."))]
            if i + j != 4:
                parts.append(up(bb(outputs[-1], f"convs.up_conv_{i+1}{j}.")))
            o = bb(bb(cat(parts), f"convs.in_conv_{i}{j}.0."), f"convs.in_conv_{i}{j}.conv_0.")
            outputs.append(o)
            hd = bb(o, f"convs.output_{i}.0.") if i != 0 else o
            # the reference recomputes output_i at every node and the last one wins (networks.py:92): only that
            # evaluation reaches the returned dict, the earlier ones are dead branches with zero gradient
            final[f"log_depth_pred_s{i}_b1hw"] = (hd, f"convs.output_{i}.1")
        prev = outputs[::-1]
    for key, (hd, name) in final.items():
        head(hd, name).accum(np.asarray(gouts[key], dtype=np.float64))
    for node in reversed(tape):
        if node.g is not None:
            node.back()
    return [n.g if n.g is not None else np.zeros_like(n.v) for n in inputs], grads


# ------------------------------------------------------- matching encoder (a16) --
# ResnetMatchingEncoder (reference modules/networks.py:149-205).  Its ResNet-18 stem/layer1 come from
# the third-party `antialiased_cnns` package (simplerecon_env.yml:20, unpinned, NOT under
# /root/reference); the functions below restate that package's published architecture
# (resnet.py / blurpool.py of adobe/antialiased-cnns: conv1 7x7/s2 -> BN -> ReLU ->
# [MaxPool2d(2, stride 1), BlurPool(filt 4, stride 2, reflect)] -> layer1 = 2 x BasicBlock(64)).
# The golden vectors for it are produced by the reference's own ResnetMatchingEncoder class built on
# oracle/refshim.py's torch.nn restatement of that backbone: the tail (networks.py:187-201) is the
# reference's code, the backbone is pinned against ATen ops only ("parity unpinned" vs the package).

def batchnorm_eval(x, sd, prefix, eps=1e-5, precision="f32"):
    """nn.BatchNorm2d in eval mode as ATen's CPU inference path computes it:
    alpha = weight / sqrt(var + eps); beta = bias - mean * alpha; y = x * alpha + beta."""
    dt, _ = _dt(precision)
    x = np.asarray(x, dtype=dt)
    w, b = np.asarray(sd[prefix + "weight"], dt), np.asarray(sd[prefix + "bias"], dt)
    m, v = np.asarray(sd[prefix + "running_mean"], dt), np.asarray(sd[prefix + "running_var"], dt)
    alpha = w * (dt(1) / np.sqrt(v + dt(eps)))
    beta = b - m * alpha
    return x * alpha[None, :, None, None] + beta[None, :, None, None]


def maxpool2_s1(x):
    """nn.MaxPool2d(kernel_size=2, stride=1): [B,C,H,W] -> [B,C,H-1,W-1]."""
    return np.maximum(np.maximum(x[:, :, :-1, :-1], x[:, :, :-1, 1:]), np.maximum(x[:, :, 1:, :-1], x[:, :, 1:, 1:]))


def blurpool4_s2(x):
    """antialiased_cnns.BlurPool(filt_size=4, stride=2, pad_type='reflect'): ReflectionPad2d((1,2,1,2)) then a
    depthwise conv with outer([1,3,3,1])/64, stride 2."""
    dt = x.dtype.type
    a = np.array([1.0, 3.0, 3.0, 1.0], dtype=x.dtype)
    f = np.outer(a, a)
    f = f / f.sum(dtype=x.dtype)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 2), (1, 2)), mode="reflect")
    Ho, Wo = (xp.shape[2] - 4) // 2 + 1, (xp.shape[3] - 4) // 2 + 1
    out = np.zeros(x.shape[:2] + (Ho, Wo), x.dtype)
    for i in range(4):
        for j in range(4):
            out += dt(f[i, j]) * xp[:, :, i:i + 2 * Ho:2, j:j + 2 * Wo:2]
    return out


def instance_norm(x, eps=1e-5, leaky=None):
    """nn.InstanceNorm2d (affine=False, no running stats): per (image, channel) biased statistics over H*W."""
    dt = x.dtype.type
    mean = x.mean(axis=(2, 3), keepdims=True, dtype=x.dtype)
    var = np.square(x - mean).mean(axis=(2, 3), keepdims=True, dtype=x.dtype)
    y = (x - mean) * (dt(1) / np.sqrt(var + dt(eps)))
    if leaky is not None:
        y = np.where(y >= 0, y, y * dt(leaky))
    return y


def resnet_block(x, sd, prefix, precision="f32"):
    """torchvision/antialiased_cnns BasicBlock, stride 1: conv3x3 -> bn -> relu -> conv3x3 -> bn -> += x -> relu."""
    t = conv2d(x, sd[prefix + "conv1.weight"], None, precision=precision)
    t = np.maximum(batchnorm_eval(t, sd, prefix + "bn1.", precision=precision), 0)
    t = conv2d(t, sd[prefix + "conv2.weight"], None, precision=precision)
    t = batchnorm_eval(t, sd, prefix + "bn2.", precision=precision) + x
    return np.maximum(t, 0)


def conv2d_replicate(x, wgt, bias, precision="f32"):
    """nn.Conv2d(k, padding=k//2, padding_mode='replicate') (networks.py:191-197)."""
    p = wgt.shape[-1] // 2
    xp = np.pad(np.asarray(x), ((0, 0), (0, 0), (p, p), (p, p)), mode="edge")
    return conv2d(xp, wgt, bias, pad=0, precision=precision)


def resnet_matching_encoder(img, sd, precision="f32", taps=None):
    """ResnetMatchingEncoder.forward (modules/networks.py:185-205), state-dict keys `net.<i>...` as nn.Sequential
    numbers them.  `taps` (dict) optionally receives the intermediate activations."""
    dt, _ = _dt(precision)
    x = conv2d(np.asarray(img, dtype=dt), sd["net.0.weight"], None, stride=2, pad=3, precision=precision)
    x = np.maximum(batchnorm_eval(x, sd, "net.1.", precision=precision), 0)
    if taps is not None:
        taps["stem"] = x
    x = blurpool4_s2(maxpool2_s1(x))
    if taps is not None:
        taps["pool"] = x
    x = resnet_block(x, sd, "net.4.0.", precision)
    x = resnet_block(x, sd, "net.4.1.", precision)
    if taps is not None:
        taps["layer1"] = x
    x = conv2d(x, sd["net.5.weight"], sd["net.5.bias"], precision=precision)
    x = instance_norm(x, leaky=0.2)
    x = conv2d_replicate(x, sd["net.8.weight"], sd["net.8.bias"], precision=precision)
    return instance_norm(x)


# ---------------------------------------------------------------- image-prior encoder (§8f "next" #1) --
# DepthModel's `encoder` is timm's tf_efficientnetv2_s feature pyramid (experiment_modules/depth_model.py:110-118:
# `timm.create_model("tf_efficientnetv2_s_in21ft1k", pretrained=True, features_only=True)`, five maps with
# 24/48/64/160/256 channels at strides 2..32).  timm is an UNPINNED third-party dependency (simplerecon_env.yml:22)
# absent from /root/reference and from this container: **PARITY UNPINNED**.  What is restated here is the published
# architecture (Tan & Le, EfficientNetV2, table 4 = timm arch definition `cn_r2_k3_s1_e1_c24_skip / er_r4_k3_s2_e4_c48 /
# er_r4_k3_s2_e4_c64 / ir_r6_k3_s2_e4_c128_se0.25 / ir_r9_k3_s1_e6_c160_se0.25 / ir_r15_k3_s2_e6_c256_se0.25`,
# stem 24, SiLU, BatchNorm eps 1e-3 and TF-"SAME" padding for the tf_* weights) with timm's EfficientNetFeatures
# state-dict names.  Sanity anchors available without timm: 19.85 M parameters in these stages (timm reports
# 21.46 M for the classifier model = + conv_head 0.33 M + bn 2.6 k + fc 1.28 M), channel list [24,48,64,160,256].

EFFNETV2_S_STAGES = (("cn", 2, 1, 1, 24), ("er", 4, 2, 4, 48), ("er", 4, 2, 4, 64),
                     ("ir", 6, 2, 4, 128), ("ir", 9, 1, 6, 160), ("ir", 15, 2, 6, 256))
EFFNETV2_S_FEATURE_STAGES = (0, 1, 2, 4, 5)


def silu(x):
    dt = x.dtype.type
    with np.errstate(over="ignore"):   # exp(-x) -> inf gives x / inf = -0, the limit
        return x / (dt(1) + np.exp(-x))


def tf_same_pad(x, k, stride):
    """Zero padding of a TensorFlow-"SAME" conv: out = ceil(in / stride), the odd pixel goes below / right."""
    def one(i):
        total = max((-(-i // stride) - 1) * stride + k - i, 0)
        return total // 2, total - total // 2
    return np.pad(np.asarray(x), ((0, 0), (0, 0), one(x.shape[2]), one(x.shape[3])))


def conv2d_same(x, wgt, stride=1, precision="f32"):
    k = wgt.shape[-1]
    return conv2d(tf_same_pad(x, k, stride), wgt, None, stride=stride, pad=0, precision=precision)


def dwconv3x3_same(x, wgt, stride=1):
    """Depthwise 3x3 conv (weight [C,1,3,3]) with TF-"SAME" padding."""
    xp = tf_same_pad(x, 3, stride)
    Ho, Wo = (xp.shape[2] - 3) // stride + 1, (xp.shape[3] - 3) // stride + 1
    w = np.asarray(wgt, dtype=x.dtype)
    out = np.zeros(x.shape[:2] + (Ho, Wo), x.dtype)
    for i in range(3):
        for j in range(3):
            out += w[None, :, 0, i, j, None, None] * xp[:, :, i:i + stride * Ho:stride, j:j + stride * Wo:stride]
    return out


def _bn_act(x, sd, prefix, act, precision):
    y = batchnorm_eval(x, sd, prefix, eps=1e-3, precision=precision)
    return silu(y) if act else y


def effnet_mbconv_block(x, sd, pre, stride, precision="f32"):
    """One MBConv block WITHOUT its identity skip: bn(conv1x1(se(silu(bn(dw3x3(silu(bn(conv1x1(x))))))))), TF-"SAME" padding
    on the depthwise conv, squeeze-excite = x * sigmoid(W2 silu(W1 mean(x) + b1) + b2), parameters under timm's names
    `pre + {conv_pw, bn1, conv_dw, bn2, se.conv_reduce, se.conv_expand, conv_pwl, bn3}`.  The one independent anchor on
    this box: transformers' EfficientNetBlock (the V1 block: same expand -> depthwise(SAME) -> SE -> project semantics)
    computes the same function (tests/test_oracle_effnet.py)."""
    dt, _ = _dt(precision)
    y = _bn_act(conv2d(x, sd[pre + "conv_pw.weight"], None, precision=precision), sd, pre + "bn1.", True, precision)
    y = _bn_act(dwconv3x3_same(y, sd[pre + "conv_dw.weight"], stride), sd, pre + "bn2.", True, precision)
    m = y.mean(axis=(2, 3), keepdims=True, dtype=y.dtype)
    w1, b1 = np.asarray(sd[pre + "se.conv_reduce.weight"], dt), np.asarray(sd[pre + "se.conv_reduce.bias"], dt)
    w2, b2 = np.asarray(sd[pre + "se.conv_expand.weight"], dt), np.asarray(sd[pre + "se.conv_expand.bias"], dt)
    h = silu(np.einsum("oc,bc->bo", w1[:, :, 0, 0], m[:, :, 0, 0]) + b1)
    g = np.einsum("oc,bc->bo", w2[:, :, 0, 0], h) + b2
    with np.errstate(over="ignore"):
        y = y * (dt(1) / (dt(1) + np.exp(-g)))[:, :, None, None]
    return _bn_act(conv2d(y, sd[pre + "conv_pwl.weight"], None, precision=precision), sd, pre + "bn3.", False, precision)


def efficientnetv2_s_features(img, sd, precision="f32", taps=None):
    """[f2, f4, f8, f16, f32] of the tf_efficientnetv2_s feature extractor for an image batch [B,3,H,W]."""
    dt, _ = _dt(precision)
    x = _bn_act(conv2d_same(np.asarray(img, dtype=dt), sd["conv_stem.weight"], 2, precision), sd, "bn1.", True,
                precision)
    feats, cin = [], 24
    for si, (kind, repeats, stride, _e, cout) in enumerate(EFFNETV2_S_STAGES):
        for bi in range(repeats):
            pre, s = f"blocks.{si}.{bi}.", (stride if bi == 0 else 1)
            skip = x if (s == 1 and cin == cout) else None
            if kind == "cn":      # ConvBnAct: act(bn(conv)) + x
                y = _bn_act(conv2d_same(x, sd[pre + "conv.weight"], s, precision), sd, pre + "bn1.", True, precision)
            elif kind == "er":    # FusedMBConv: bn(conv1x1(act(bn(conv3x3)))) + x
                y = _bn_act(conv2d_same(x, sd[pre + "conv_exp.weight"], s, precision), sd, pre + "bn1.", True,
                            precision)
                y = _bn_act(conv2d(y, sd[pre + "conv_pwl.weight"], None, precision=precision), sd, pre + "bn2.",
                            False, precision)
            else:                 # MBConv: expand 1x1, depthwise 3x3, squeeze-excite, project 1x1
                y = effnet_mbconv_block(x, sd, pre, s, precision)
            x = y + skip if skip is not None else y
            cin = cout
            if taps is not None:
                taps[pre[:-1]] = x
        if si in EFFNETV2_S_FEATURE_STAGES:
            feats.append(x)
    return feats


# ---------------------------------------------------------------- TSDF fusion (§8f "next" #2) --
# Restates tools/tsdf.py (TSDF.from_bounds :69-97, generate_voxel_coords :99-111, TSDFFuser.project_to_camera
# :218-236, integrate_depth :238-320) as the reference executes it: EVERY tensor is fp16 (OurFuser.fuse_frames
# passes depth / K / cam_T_world through .half(), fusers_helper.py:62-68) and every torch op on fp16 tensors
# computes in fp32 and rounds its result to fp16.  numpy's float16 ufuncs do exactly that, so the restatement is
# written op for op on float16 arrays; the two matmuls round once after an fp32 dot product (products of two
# halves are exact in fp32).  Checked bit-for-bit against the reference class run on CPU (tests/golden).

_F16 = np.float16


def _hs(s):
    """python scalar -> fp16 the way torch converts it (c10::Half(float(double)): through fp32)."""
    return np.float16(np.float32(s))


def tsdf_from_bounds(bounds, voxel_size, vox_mod=8):
    """TSDF.from_bounds: (origin float32 [3], dims (X, Y, Z), voxel_coords float16 [3,X,Y,Z], values, weights)."""
    dims = tuple(int(np.ceil((bounds[a + "max"] - bounds[a + "min"]) / voxel_size / vox_mod)) * vox_mod for a in "xyz")
    origin = np.array([bounds["xmin"], bounds["ymin"], bounds["zmin"]], np.float32)
    grid = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), 0)
    # int64 grid * python float -> float32 tensor product with the scalar cast to float32 (generate_voxel_coords :108)
    coords = (origin.reshape(3, 1, 1, 1) + grid.astype(np.float32) * np.float32(voxel_size)).astype(_F16)
    values = -np.ones(dims, _F16)
    weights = np.zeros(dims, _F16)
    return origin, dims, coords, values, weights


def _half_matmul(a, b):
    """torch.matmul on fp16 operands: exact fp32 products, fp32 accumulation in index order, one rounding to fp16."""
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    acc = a32[:, 0:1] * b32[0:1]
    for k in range(1, a32.shape[1]):
        acc = acc + a32[:, k:k + 1] * b32[k:k + 1]
    return acc.astype(_F16)


def _half_div_scalar(a, s):
    """fp16 tensor / python scalar on torch's CPU path: fp32 division by the scalar AS FLOAT32 (not rounded to fp16),
    one rounding of the quotient to fp16 (probed on torch 2.10; tensor +- scalar and comparisons DO round the scalar
    to fp16 first)."""
    return (a.astype(np.float32) / np.float32(s)).astype(_F16)


def tsdf_integrate(values, weights, coords, depth_b1hw, cam_T_world_b44, K_b44, min_depth=0.5, max_depth=5.0,
                   voxel_size=0.04, truncation_size=3.0, maxW=100.0, depth_mask_b1hw=None):
    """TSDFFuser.integrate_depth: updates `values` / `weights` ([X,Y,Z] float16) in place, frame after frame."""
    depth_b1hw = np.asarray(depth_b1hw, _F16)
    T, K = np.asarray(cam_T_world_b44, _F16), np.asarray(K_b44, _F16)
    B, _, H, W = depth_b1hw.shape
    trunc = truncation_size * voxel_size                       # python float (tsdf.py:214-216)
    X = coords.reshape(3, -1)
    hom = np.concatenate([X, np.ones((1, X.shape[1]), _F16)], 0)
    vals, wts = values.reshape(-1), weights.reshape(-1)
    for b in range(B):
        depth = depth_b1hw[b, 0]
        if depth_mask_b1hw is not None:
            depth = depth.copy()
            depth[~np.asarray(depth_mask_b1hw[b, 0], bool)] = _F16(-1)
        P = _half_matmul(K[b], T[b])[:3]                       # world_to_pix_P_b34 (:226)
        cam = _half_matmul(P, hom)                             # cam_points_b3N (:232)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            z = cam[2]
            px, py = cam[0] / z, cam[1] / z                    # (:233)
            gx = _F16(2) * px / _F16(W) - _F16(1)              # 2 * pix / img_size - 1 (:268)
            gy = _F16(2) * py / _F16(H) - _F16(1)
            # F.grid_sample(mode="nearest", padding_mode="zeros", align_corners=False) on Half: ATen's
            # grid_sampler_compute_source_index in Half arithmetic, then nearbyint (:275-279)
            ix = ((gx + _F16(1)) * _F16(W) - _F16(1)) / _F16(2)
            iy = ((gy + _F16(1)) * _F16(H) - _F16(1)) / _F16(2)
            # a non-finite Half coordinate (|pix| overflows fp16 for voxels next to the camera plane) indexes
            # texel 0 on ATen's CPU path (probed: inf / -inf / nan -> column or row 0), it is not "out of bounds"
            ixn = np.where(np.isfinite(ix), np.rint(ix.astype(np.float32)), np.float32(0))
            iyn = np.where(np.isfinite(iy), np.rint(iy.astype(np.float32)), np.float32(0))
            inb = (ixn >= 0) & (ixn < W) & (iyn >= 0) & (iyn < H)
            sd = np.zeros(X.shape[1], _F16)
            sd[inb] = depth[iyn[inb].astype(np.int64), ixn[inb].astype(np.int64)]
            conf = np.clip(_F16(1.0) - _half_div_scalar(sd - _hs(min_depth), max_depth - min_depth), _F16(0), _F16(1))
            conf = (conf * conf).astype(_F16)                  # ** 2 (:283-285)
            dist = sd - z
            tv = np.clip(_half_div_scalar(dist, trunc), _F16(-1), _F16(1))
            valid = (z > 0) & (dist > _hs(-trunc)) & (sd > 0) & (z < _hs(max_depth)) & (conf > 0)
        ov, ow = vals[valid], wts[valid]
        nv, cf = tv[valid], conf[valid]
        rate = np.where(cf < ow, _F16(2), _F16(5)).astype(_F16)
        nw = _half_div_scalar(cf * rate, maxW)
        tw = ow + nw
        vals[valid] = (ov * ow + nv * nw) / tw
        wts[valid] = np.minimum(tw, _F16(1))
    return values, weights
