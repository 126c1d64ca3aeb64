"""Conv building blocks -- API/state-dict compatible with reference modules/layers.py.

`BasicBlock` keeps the reference's sub-module names (conv1, bn1, relu, conv2, bn2, downsample)
so checkpoints load unchanged; its forward runs hand-written gfx950 implicit-GEMM kernels
(fp32 MFMA, fused bias + residual + LeakyReLU epilogue) through simplerecon_amd.ops."""
from typing import Callable, Optional

import torch.nn as nn
from torch import Tensor


def conv3x3(in_planes: int, out_planes: int, stride: int = 1, groups: int = 1, dilation: int = 1,
            bias: bool = False) -> nn.Conv2d:
    """3x3 convolution with padding (reference layers.py:7-17); parameter holder."""
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups,
                     bias=bias, dilation=dilation)


def conv1x1(in_planes: int, out_planes: int, stride: int = 1, bias: bool = False) -> nn.Conv2d:
    """1x1 convolution (reference layers.py:20-22); parameter holder."""
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=bias)


class BasicBlock(nn.Module):
    """conv3x3(s) -> LeakyReLU(0.2) -> conv3x3 -> + skip -> LeakyReLU(0.2)  (reference layers.py:24-85).

    With the reference's default `norm_layer=nn.Identity` the convs carry a bias and there is no
    normalisation; that is the only configuration SimpleRecon uses and the only one the HIP path
    implements."""
    expansion: int = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = nn.Identity) -> None:
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        bias = norm_layer == nn.Identity
        self.conv1 = conv3x3(inplanes, planes, stride, bias=bias)
        self.bn1 = norm_layer(planes)
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.conv2 = conv3x3(planes, planes, bias=bias)
        self.bn2 = norm_layer(planes)
        if inplanes == planes * self.expansion and stride == 1:
            self.downsample = None
        else:
            conv = conv1x1 if stride == 1 else conv3x3
            self.downsample = nn.Sequential(conv(inplanes, planes * self.expansion, bias=bias, stride=stride),
                                            norm_layer(planes * self.expansion))
        self.stride = stride

    def forward(self, x: Tensor, out: Optional[Tensor] = None) -> Tensor:
        """x: [B,C,H,W] (any memory format; channels_last avoids a repack).  `out` optionally names
        a preallocated (channel-slice of a) channels_last tensor to write into."""
        from . import ops
        return ops.basic_block(self, x, out=out)


class TensorFormatter(nn.Module):
    """B x M x C x H x W <-> (B*M) x C x H x W reshaper (reference layers.py:87-121)."""

    def __init__(self):
        super().__init__()
        self.batch_size = None
        self.depth_chns = None

    def _expand_batch_with_channels(self, x):
        if x.dim() != 5:
            raise ValueError("TensorFormatter expects tensors with 5 dimensions, not {}!".format(len(x.shape)))
        self.batch_size, self.depth_chns, chns, height, width = x.shape
        return x.view(self.batch_size * self.depth_chns, chns, height, width)

    def _reduce_batch_to_channels(self, x):
        if self.batch_size is None or self.depth_chns is None:
            raise ValueError("Cannot  call _reduce_batch_to_channels without first calling"
                             "_expand_batch_with_channels!")
        _, chns, height, width = x.shape
        return x.view(self.batch_size, self.depth_chns, chns, height, width)

    def forward(self, x, apply_func):
        x = self._expand_batch_with_channels(x)
        x = apply_func(x)
        return self._reduce_batch_to_channels(x)
