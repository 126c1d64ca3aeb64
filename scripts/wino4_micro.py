"""F(4x4, 3x3) vs F(2x2, 3x3) Winograd on the full-resolution layer shapes (HIP events, both kernels in one process), with the
error of each against an fp64 ATen convolution."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_wino4 import _run, _ref64
from simplerecon_amd import ops
dev = "cuda:0"
shapes = [(8, 64, 240, 320, 64), (8, 192, 240, 320, 64), (8, 128, 240, 320, 64), (8, 24, 240, 320, 64), (1, 64, 240, 320, 64),
          (8, 64, 120, 160, 64), (4, 64, 368, 480, 64)]
if os.environ.get("SR_MICRO_SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SR_MICRO_SHAPES"].split(",")]
for (B, ci, H, W, co) in shapes:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.empty_nhwc(B, co, H, W, dev)
    line = f"{str((B, ci, H, W, co)):26s}"
    with torch.inference_mode():
        ref = _ref64(x[:1], conv, res[:1], 0.2)
        for kind in ("w2", "w4", "w4_ws"):
            f = lambda: _run(kind, x, conv, res, 0.2, out=out)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n): f()
            e1.record(); e1.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / n
            err = (out[:1].double() - ref).abs().max().item() / ref.abs().max().item()
            fl = 2.0 * B * H * W * co * ci * 9
            mult = 16 / 36 if kind == "w2" else 36 / 144
            line += f"  {kind}: {t * 1e6:7.1f} us util {fl * mult / t / 1e12 / 157.3:5.3f} err {err:.0e}"
    print(line, flush=True)
