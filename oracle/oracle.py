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
