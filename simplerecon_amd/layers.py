"""Conv building blocks with the public surface of the reference's modules/layers.py (`conv3x3`, `conv1x1`,
`BasicBlock`, `TensorFormatter`): same constructor arguments, same sub-module names (conv1, bn1, relu, conv2, bn2,
downsample.0) so that reference checkpoints load unchanged.  The modules only HOLD parameters; every forward runs the
hand-written gfx950 kernels (fp32 MFMA implicit-GEMM / Winograd convolutions with fused bias + residual + LeakyReLU
epilogue) through simplerecon_amd.ops."""
from typing import Callable, Optional

import torch.nn as nn
from torch import Tensor


def _conv_holder(cin: int, cout: int, k: int, stride: int, bias: bool, dilation: int = 1, groups: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=dilation * (k // 2), dilation=dilation,
                     groups=groups, bias=bias)


def conv3x3(in_planes: int, out_planes: int, stride: int = 1, groups: int = 1, dilation: int = 1,
            bias: bool = False) -> nn.Conv2d:
    """Parameter holder of a padded 3x3 convolution (reference layers.py:7-17)."""
    return _conv_holder(in_planes, out_planes, 3, stride, bias, dilation, groups)


def conv1x1(in_planes: int, out_planes: int, stride: int = 1, bias: bool = False) -> nn.Conv2d:
    """Parameter holder of a 1x1 convolution (reference layers.py:20-22)."""
    return _conv_holder(in_planes, out_planes, 1, stride, bias)


class BasicBlock(nn.Module):
    """act(conv2(act(conv1(x))) + skip(x)) with act = LeakyReLU(0.2); conv1 carries the stride; skip is the identity,
    a 1x1 conv when only the channel count changes, or a strided 3x3 conv when the block downsamples
    (reference layers.py:24-85).

    SimpleRecon always builds it with `norm_layer=nn.Identity` (biased convs, no normalisation): the only configuration
    the HIP path executes -- others can be constructed (for state-dict inspection) but refuse to run."""
    expansion: int = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = nn.Identity) -> None:
        super().__init__()
        if (groups, base_width) != (1, 64):
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        norm = nn.BatchNorm2d if norm_layer is None else norm_layer
        biased = norm is nn.Identity
        width = planes * self.expansion
        self.stride = stride
        self.conv1, self.bn1 = conv3x3(inplanes, planes, stride, bias=biased), norm(planes)
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.conv2, self.bn2 = conv3x3(planes, planes, bias=biased), norm(planes)
        self.downsample = None
        if stride != 1 or inplanes != width:
            skip = conv3x3(inplanes, width, stride, bias=biased) if stride != 1 else conv1x1(inplanes, width, bias=biased)
            self.downsample = nn.Sequential(skip, norm(width))

    def forward(self, x: Tensor, out: Optional[Tensor] = None) -> Tensor:
        """x: [B,C,H,W] (any memory format; channels_last avoids a repack).  `out` optionally names a preallocated
        (channel slice of a) channels_last tensor to write into, e.g. the consumer's concat buffer."""
        from . import autograd_ops, ops
        if autograd_ops.grad_wanted(x, self):   # training path: differentiable operators (HIP forward and backward)
            y = autograd_ops.basic_block(self, x)
            if out is not None:
                out.copy_(y)                    # (autograd tracks the slice write; the training forwards avoid `out=`)
                return out
            return y
        return ops.basic_block(self, x, out=out)


class TensorFormatter(nn.Module):
    """Applies a per-image function to a [B, M, C, H, W] stack by folding M into the batch and unfolding the result
    again (reference layers.py:87-121; used for the source-view images / features).  Like the reference's it
    remembers the last fold (`batch_size`, `depth_chns`) so the two halves can also be called separately."""

    def __init__(self):
        super().__init__()
        self.batch_size = None
        self.depth_chns = None

    def _expand_batch_with_channels(self, x):
        if x.dim() != 5:
            raise ValueError(f"TensorFormatter expects tensors with 5 dimensions, not {x.dim()}!")
        self.batch_size, self.depth_chns = int(x.shape[0]), int(x.shape[1])
        return x.flatten(0, 1) if x.is_contiguous() else x.reshape(-1, *x.shape[2:])

    def _reduce_batch_to_channels(self, x):
        if None in (self.batch_size, self.depth_chns):
            raise ValueError("Cannot call _reduce_batch_to_channels without first calling "
                             "_expand_batch_with_channels!")
        return x.unflatten(0, (self.batch_size, self.depth_chns))

    def forward(self, x, apply_func):
        return self._reduce_batch_to_channels(apply_func(self._expand_batch_with_channels(x)))
