"""torch.nn.functional (ATen, CPU) restatement of the tf_efficientnetv2_s feature extractor -- a second, independent
statement of the same PUBLIC architecture, used only to cross-check oracle/oracle.py::efficientnetv2_s_features on
CPU.  It is not timm (absent here), so it does not pin parity against the reference's dependency: see DESIGN.md §3.7."""
import math

import torch
import torch.nn.functional as F

ARCH = (("cn", 2, 1, 1, 24), ("er", 4, 2, 4, 48), ("er", 4, 2, 4, 64), ("ir", 6, 2, 4, 128), ("ir", 9, 1, 6, 160),
        ("ir", 15, 2, 6, 256))


def _same(x, k, s):
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))


def _conv_same(x, w, s=1, groups=1):
    return F.conv2d(_same(x, w.shape[-1], s), w, None, stride=s, groups=groups)


_TRAINING = False


def _bn(x, sd, pre, act):
    rm, rv = sd[pre + "running_mean"], sd[pre + "running_var"]
    if _TRAINING:   # batch statistics; the running buffers are updated on clones (callers compare gradients / outputs)
        rm, rv = rm.detach().clone(), rv.detach().clone()
    y = F.batch_norm(x, rm, rv, sd[pre + "weight"], sd[pre + "bias"], training=_TRAINING, eps=1e-3)
    return F.silu(y) if act else y


def features(img, sd, training=False):
    """training=True: BatchNorm with batch statistics (what the reference's train.py runs); differentiable either way."""
    global _TRAINING
    _TRAINING = training
    x = _bn(_conv_same(img, sd["conv_stem.weight"], 2), sd, "bn1.", True)
    feats, cin = [], 24
    for si, (kind, reps, stride, _e, cout) in enumerate(ARCH):
        for bi in range(reps):
            pre, s = f"blocks.{si}.{bi}.", (stride if bi == 0 else 1)
            skip = s == 1 and cin == cout
            if kind == "cn":
                y = _bn(_conv_same(x, sd[pre + "conv.weight"], s), sd, pre + "bn1.", True)
            elif kind == "er":
                y = _bn(_conv_same(x, sd[pre + "conv_exp.weight"], s), sd, pre + "bn1.", True)
                y = _bn(F.conv2d(y, sd[pre + "conv_pwl.weight"]), sd, pre + "bn2.", False)
            else:
                y = _bn(F.conv2d(x, sd[pre + "conv_pw.weight"]), sd, pre + "bn1.", True)
                y = _bn(_conv_same(y, sd[pre + "conv_dw.weight"], s, groups=y.shape[1]), sd, pre + "bn2.", True)
                g = y.mean((2, 3), keepdim=True)
                g = F.silu(F.conv2d(g, sd[pre + "se.conv_reduce.weight"], sd[pre + "se.conv_reduce.bias"]))
                g = F.conv2d(g, sd[pre + "se.conv_expand.weight"], sd[pre + "se.conv_expand.bias"])
                y = y * torch.sigmoid(g)
                y = _bn(F.conv2d(y, sd[pre + "conv_pwl.weight"]), sd, pre + "bn3.", False)
            x = y + x if skip else y
            cin = cout
        if si in (0, 1, 2, 4, 5):
            feats.append(x)
    return feats
