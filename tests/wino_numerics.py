"""Input families and error measures for the Winograd numerics study: the F(4x4, 3x3) kernel's error constant
is 5-8x F(2x2)'s on zero-mean inputs and Winograd error grows with the DC offset / dynamic range of the input -- what post-ReLU
maps (reference modules/networks.py:176-182, the matching encoder's layer1) and trained decoders produce.  Shared by
tests/test_gpu_wino4.py (bounds) and scripts/wino4_numerics.py (the table in profiles/r06_wino4_numerics.txt)."""
import torch

# name -> (input generator, weight gain).  Every generator returns a [B, C, H, W] fp32 tensor (not yet channels-last).
FAMILIES = ("randn", "relu", "relu_dc3", "dc10", "dc100", "tails", "tails_relu", "gain8", "smooth")


def make_input(family, shape, device, seed=0):
    g = torch.Generator(device=device)
    g.manual_seed(1000 + seed)
    b, c, h, w = shape
    x = torch.randn(shape, device=device, generator=g)
    if family in ("randn", "gain8"):
        return x
    if family == "relu":            # |N(0,1)|: mean 0.80, std 0.60 -> mean / std = 1.3 (a post-ReLU map)
        return x.abs()
    if family == "relu_dc3":        # post-ReLU map riding on a bias: mean / std = 6
        return x.abs() + 3.0
    if family == "dc10":
        return x + 10.0
    if family == "dc100":
        return x + 100.0
    if family in ("tails", "tails_relu"):   # about one +-1e3 outlier per 16 x 16 region and channel group of 16
        m = torch.rand(shape, device=device, generator=g) < 1.0 / (256 * 16)
        s = torch.where(torch.rand(shape, device=device, generator=g) < 0.5, -1.0, 1.0)
        base = x.abs() if family == "tails_relu" else x
        return torch.where(m, 1.0e3 * s * (1.0 + x.abs()), base)
    if family == "smooth":          # a smooth ramp (image-like low frequencies) plus small noise: neighbouring taps nearly cancel
        yy = torch.linspace(0, 3.0, h, device=device).view(1, 1, h, 1)
        xx = torch.linspace(0, 4.0, w, device=device).view(1, 1, 1, w)
        ph = torch.rand((1, c, 1, 1), device=device, generator=g) * 6.28
        return 5.0 * torch.sin(yy + ph) * torch.cos(xx - ph) + 5.0 + 0.05 * x
    raise ValueError(family)


def weight_gain(family):
    return 8.0 if family == "gain8" else 1.0


def errors(y, ref, local_scale=None):
    """(range-relative max error, rms-relative error, max error relative to the per-pixel magnitude bound sum |w||x| + |b| + |r|)."""
    d = (y.double() - ref).abs()
    rng = d.max().item() / ref.abs().max().item()
    rms = (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    loc = (d / local_scale).max().item() if local_scale is not None else float("nan")
    return rng, rms, loc


def local_scale64(x, conv, res):
    """sum |w| |x| + |bias| + |residual| per output: the magnitude a direct fp32 convolution's error is proportional to."""
    s = torch.nn.functional.conv2d(x.double().abs(), conv.weight.double().abs(),
                                   conv.bias.double().abs() if conv.bias is not None else None, padding=1)
    if res is not None:
        s = s + res.double().abs()
    return s.clamp_min(1e-30)
