"""Split-precision Winograd convolution (SR_WINO_SPLIT=bf16|f16, fenced experiment) next to the fp32-MFMA kernel: error of
each against an fp64 convolution, and the launch time, on layer shapes of the hero conv stack + ragged / border shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from simplerecon_amd import ops

dev = "cuda:0"
MODES = [("fp32", "0"), ("bf16", "bf16"), ("f16", "f16")]
shapes = [(8, 64, 240, 320, 64), (8, 192, 240, 320, 64), (8, 64, 120, 160, 64), (8, 128, 60, 80, 128), (8, 256, 30, 40, 256),
          (8, 384, 15, 20, 384), (1, 64, 240, 320, 64), (1, 384, 15, 20, 384), (2, 24, 61, 83, 40), (1, 20, 37, 50, 32)]
if os.environ.get("SR_MICRO_SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SR_MICRO_SHAPES"].split(",")]
torch.manual_seed(0)
for (B, ci, H, W, co) in shapes:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    ref = F.leaky_relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double(), 0.2)
    rng = float(ref.abs().max())
    line = f"{str((B, ci, H, W, co)):26s}"
    for name, v in MODES:
        _lib.set_option("SR_WINO_SPLIT", v)
        out = ops.empty_nhwc(B, co, H, W, dev)
        with torch.inference_mode():
            f = lambda: ops.conv2d(x, conv, residual=res, leaky=0.2, out=out)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / n
        err = (out.double() - ref).abs()
        line += f"  {name}: {t*1e6:7.1f} us  max {float(err.max())/rng:.2e} rms {float((err**2).mean().sqrt())/rng:.2e}"
    _lib.set_option("SR_WINO_SPLIT", 0)
    print(line, flush=True)
