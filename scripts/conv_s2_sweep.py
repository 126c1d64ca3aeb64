"""The stride-2 3x3 convolutions of the step (sr_conv_kernel<3, 2, ...>) under every forced tile plan (SR_CONV_TILE = shape + 10 nt:
shape 1 = 4x32 pixels, 2 = 16x8; nt = 32-channel tiles per wave), HIP events over back-to-back launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops, _lib
dev = "cuda:0"
shapes = [(8, 24, 240, 320, 96), (8, 48, 120, 160, 192), (8, 64, 120, 160, 128), (8, 128, 60, 80, 256), (8, 256, 30, 40, 384),
          (1, 64, 120, 160, 128), (1, 128, 60, 80, 256), (1, 256, 30, 40, 384)]
for (B, ci, H, W, co) in shapes:
    conv = torch.nn.Conv2d(ci, co, 3, stride=2, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    line = f"{str((B, ci, H, W, co)):28s}"
    ref = None
    for plan in (0, 11, 21, 12, 22):
        _lib.set_option("SR_CONV_TILE", plan)
        with torch.inference_mode():
            f = lambda: ops.conv2d(x, conv, leaky=0.2)
            for _ in range(3): y = f()
            torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n): f()
            e1.record(); e1.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / n
        fl = 2.0 * B * (H // 2) * (W // 2) * co * ci * 9
        line += f"  plan {plan:2d}: {t * 1e6:7.1f} us {fl / t / 1e12:5.1f} TF{'' if torch.equal(y, ref) else ' (differs)'}"
    _lib.set_option("SR_CONV_TILE", 0)
    print(line, flush=True)
