"""Image-prior encoder of DepthModel: the EfficientNetV2-S feature pyramid, on the hand-written gfx950 kernels.

The reference builds it with `timm.create_model("tf_efficientnetv2_s_in21ft1k", pretrained=True,
features_only=True)` (reference experiment_modules/depth_model.py:110-116) and reads five maps at strides
2 / 4 / 8 / 16 / 32 with 24 / 48 / 64 / 160 / 256 channels (`feature_info.channels()`, :118).  timm is a third-party
dependency that is neither vendored by the reference nor installed here, so this module restates the PUBLIC
architecture (Tan & Le, "EfficientNetV2", table 4; timm's `efficientnetv2_s` arch definition
  cn_r2_k3_s1_e1_c24_skip | er_r4_k3_s2_e4_c48 | er_r4_k3_s2_e4_c64 | ir_r6_k3_s2_e4_c128_se0.25 |
  ir_r9_k3_s1_e6_c160_se0.25 | ir_r15_k3_s2_e6_c256_se0.25
with the `tf_` conventions: BatchNorm eps 1e-3 and TensorFlow-"SAME" padding, i.e. stride-2 convs pad 0 above/left and
1 below/right on even-sized maps).  Parameter / buffer names follow timm's `EfficientNetFeatures` (conv_stem, bn1,
blocks.<stage>.<i>.{conv, conv_exp, conv_pw, conv_dw, conv_pwl, bn1-3, se.conv_reduce, se.conv_expand}) so that the
`encoder.*` entries of a reference checkpoint load with load_state_dict.  PARITY UNPINNED against timm itself (no
source, no weights in this container): tests compare the HIP path with oracle/ and a torch restatement of the same
public definition.

The modules only hold parameters.  Every forward runs HIP kernels (inference: ops.conv2d with folded eval-mode
BatchNorm and SiLU epilogue, ops.dwconv3x3, ops.se_scale_, ops.add_; training -- a gradient is wanted or a BatchNorm
layer is in training mode: the differentiable operators of train_ops, forward and backward on HIP kernels); there is no
torch fallback."""
import os
from typing import List

import torch
from torch import nn

from . import ops
from . import train_ops as T


def _tf_pads(x, conv):
    """TF-"SAME" pads of a 3x3 conv for this input (None = the symmetric pad 1)."""
    pads = ops.tf_same_pads(x.shape[2], x.shape[3], conv.kernel_size[0], conv.stride[0])
    return None if pads == (1, 1, 1, 1) else pads

BN_EPS = 1e-3          # tf_* models
FUSE_SE_GATE = os.environ.get("SR_SE_GATE_FUSED", "1") != "0"   # 0: separate in-place scaling pass (r03)
# 1: expansion -> depthwise -> squeeze-excite gates in ONE launch (csrc/sr_mbconv_fused.hip, r05).  OFF by default: measured
# 47 / 80-90 / 93 us per block (stages 3 / 4 / 5, batch 8) against 52 / 78 / 53 us for the launch-per-operator path -- its
# 16-channel slices re-read the block input 32-96 times from L2 and write half cache lines
FUSE_MBCONV = os.environ.get("SR_MBCONV_FUSED", "0") == "1"
STEM_CHANNELS = 24
# (block type, repeats, stride, expansion, output channels, squeeze-excite ratio w.r.t. the block input)
STAGES = (("cn", 2, 1, 1, 24, 0.0), ("er", 4, 2, 4, 48, 0.0), ("er", 4, 2, 4, 64, 0.0),
          ("ir", 6, 2, 4, 128, 0.25), ("ir", 9, 1, 6, 160, 0.25), ("ir", 15, 2, 6, 256, 0.25))
FEATURE_STAGES = (0, 1, 2, 4, 5)   # the last stage at each stride: reductions 2, 4, 8, 16, 32


def _conv(cin, cout, k, stride=1, groups=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=bias)


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS)


class ConvBnAct(nn.Module):
    """3x3 conv + BN + SiLU with identity skip (stage 0)."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv, self.bn1 = _conv(cin, cout, 3, stride), _bn(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        y = ops.conv2d(x, self.conv, bn=self.bn1, act="silu", tf_same=True)
        return ops.add_(y, x) if self.has_skip else y   # the sum follows the activation: not a conv epilogue

    def forward_train(self, x):
        y = T.batch_norm_act(T.conv(x, self.conv, pads=_tf_pads(x, self.conv)), self.bn1, act=T.ACT_SILU)
        return T.add(y, x) if self.has_skip else y


class EdgeResidual(nn.Module):
    """FusedMBConv: 3x3 expansion conv + BN + SiLU, 1x1 projection + BN, identity skip (stages 1-2)."""

    def __init__(self, cin, cout, stride, expansion):
        super().__init__()
        mid = cin * expansion
        self.conv_exp, self.bn1 = _conv(cin, mid, 3, stride), _bn(mid)
        self.se = nn.Identity()
        self.conv_pwl, self.bn2 = _conv(mid, cout, 1), _bn(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        t = ops.conv2d(x, self.conv_exp, bn=self.bn1, act="silu", tf_same=True)
        return ops.conv2d(t, self.conv_pwl, bn=self.bn2, residual=x if self.has_skip else None, library_gemm=True)

    def forward_train(self, x):
        t = T.batch_norm_act(T.conv(x, self.conv_exp, pads=_tf_pads(x, self.conv_exp)), self.bn1, act=T.ACT_SILU)
        y = T.batch_norm_act(T.conv(t, self.conv_pwl), self.bn2)
        return T.add(y, x) if self.has_skip else y


class SqueezeExcite(nn.Module):
    def __init__(self, channels, reduced):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, reduced, 1, bias=True)
        self.conv_expand = nn.Conv2d(reduced, channels, 1, bias=True)


class InvertedResidual(nn.Module):
    """MBConv: 1x1 expansion + BN + SiLU, depthwise 3x3 + BN + SiLU, squeeze-excite, 1x1 projection + BN, skip."""

    def __init__(self, cin, cout, stride, expansion, se_ratio):
        super().__init__()
        mid = cin * expansion
        self.conv_pw, self.bn1 = _conv(cin, mid, 1), _bn(mid)
        self.conv_dw, self.bn2 = _conv(mid, mid, 3, stride, groups=mid), _bn(mid)
        self.se = SqueezeExcite(mid, int(round(cin * se_ratio)))
        self.conv_pwl, self.bn3 = _conv(mid, cout, 1), _bn(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        if FUSE_MBCONV and ops.USE_PW_1X1 and FUSE_SE_GATE and ops.mbconv_fused_supported(x, self.conv_pw, self.conv_dw, self.se):
            # r05: expansion -> depthwise -> squeeze-excite gates in ONE launch (csrc/sr_mbconv_fused.hip), then the projection
            d, gate = ops.mbconv_expand_dw_se(x, self.conv_pw, self.bn1, self.conv_dw, self.bn2, self.se)
            return ops.conv2d(d, self.conv_pwl, bn=self.bn3, residual=x if self.has_skip else None, gate=gate)
        t = ops.conv2d(x, self.conv_pw, bn=self.bn1, act="silu", library_gemm=True)
        d, pool = ops.dwconv3x3(t, self.conv_dw, bn=self.bn2, act="silu", tf_same=True, want_pool=True)
        if ops.USE_PW_1X1 and FUSE_SE_GATE:
            # the squeeze-excite gate rides in the projection's A operand: the gated map is never written
            gate = ops.se_gates(pool, d.shape[2] * d.shape[3], self.se.conv_reduce, self.se.conv_expand)
            return ops.conv2d(d, self.conv_pwl, bn=self.bn3, residual=x if self.has_skip else None, gate=gate)
        ops.se_scale_(d, pool, self.se.conv_reduce, self.se.conv_expand)
        return ops.conv2d(d, self.conv_pwl, bn=self.bn3, residual=x if self.has_skip else None, library_gemm=True)

    def forward_train(self, x):
        t = T.batch_norm_act(T.conv(x, self.conv_pw), self.bn1, act=T.ACT_SILU)
        pads = ops.tf_same_pads(t.shape[2], t.shape[3], 3, self.conv_dw.stride[0])
        d = T.batch_norm_act(T.dwconv3x3(t, self.conv_dw, pads), self.bn2, act=T.ACT_SILU)
        d = T.squeeze_excite(d, self.se.conv_reduce, self.se.conv_expand)
        y = T.batch_norm_act(T.conv(d, self.conv_pwl), self.bn3)
        return T.add(y, x) if self.has_skip else y


class _FeatureInfo:
    def __init__(self, chans, reductions):
        self._chans, self._red = list(chans), list(reductions)

    def channels(self):
        return list(self._chans)

    def reduction(self):
        return list(self._red)


class EfficientNetV2SFeatures(nn.Module):
    """`encoder(image [B,3,H,W]) -> [f2, f4, f8, f16, f32]` with 24 / 48 / 64 / 160 / 256 channels (channels-last
    views), the interface DepthModel.forward expects from its `encoder` (reference depth_model.py:358)."""

    def __init__(self):
        super().__init__()
        self.conv_stem, self.bn1 = _conv(3, STEM_CHANNELS, 3, 2), _bn(STEM_CHANNELS)
        stages, cin = [], STEM_CHANNELS
        for kind, repeats, stride, expansion, cout, se_ratio in STAGES:
            blocks = []
            for i in range(repeats):
                s = stride if i == 0 else 1
                if kind == "cn":
                    blocks.append(ConvBnAct(cin, cout, s))
                elif kind == "er":
                    blocks.append(EdgeResidual(cin, cout, s, expansion))
                else:
                    blocks.append(InvertedResidual(cin, cout, s, expansion, se_ratio))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.num_ch_enc = [STAGES[i][4] for i in FEATURE_STAGES]
        self.feature_info = _FeatureInfo(self.num_ch_enc, [2, 4, 8, 16, 32])
        self.eval()

    def _train_path(self, image):
        from . import autograd_ops
        return autograd_ops.grad_wanted([image], self) or autograd_ops.any_batchnorm_training(self)

    def forward(self, image: torch.Tensor, on_level=None) -> List[torch.Tensor]:
        """`on_level(x)`: called right behind the launch that produced a pyramid level (DepthModel records a HIP event there)."""
        if self._train_path(image):
            # training: the same graph on the differentiable operators of train_ops (conv -> BatchNorm per its own mode ->
            # SiLU unfused; drop-path rate 0 like timm's default for this model)
            x = T.batch_norm_act(T.conv(image, self.conv_stem, pads=_tf_pads(image, self.conv_stem)), self.bn1,
                                 act=T.ACT_SILU)
            feats = []
            for i, stage in enumerate(self.blocks):
                for blk in stage:
                    x = blk.forward_train(x)
                if i in FEATURE_STAGES:
                    feats.append(x)
            return feats
        x = ops.rgb_stem3x3s2(image, self.conv_stem, bn=self.bn1, act="silu", tf_same=True)
        feats = []
        for i, stage in enumerate(self.blocks):
            x = stage(x)
            if i in FEATURE_STAGES:
                feats.append(x)
                if on_level is not None:
                    on_level(x)
        return feats
