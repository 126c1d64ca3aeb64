"""Winograd layer shapes of the hero conv stack: 4-wave kernel (SR_WINO8=0) vs 8-wave kernel (SR_WINO8=2), HIP events.
SR_WINO8 is read per call, so both run in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops, _lib
dev = "cuda:0"
shapes = [(8, 64, 240, 320, 64), (8, 192, 240, 320, 64), (8, 128, 240, 320, 64), (8, 64, 120, 160, 64), (8, 192, 120, 160, 64),
          (8, 128, 60, 80, 128), (8, 256, 30, 40, 256), (8, 384, 15, 20, 384), (8, 24, 240, 320, 64), (1, 64, 240, 320, 64),
          (1, 64, 120, 160, 64), (1, 128, 60, 80, 128)]
if os.environ.get("SR_MICRO_SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SR_MICRO_SHAPES"].split(",")]
modes = os.environ.get("SR_MICRO_MODES", "0,2").split(",")
lib = _lib.lib()
for (B, ci, H, W, co) in shapes:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.empty_nhwc(B, co, H, W, dev)
    ts, ys = [], []
    for mode in modes:
        os.environ["SR_WINO8"] = mode
        with torch.inference_mode():
            f = lambda: ops.conv2d(x, conv, residual=res, leaky=0.2, out=out)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n): f()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3 / n)
            ys.append(out.clone())
    fl = 2.0 * B * H * W * co * ci * 9
    same = all(torch.equal(ys[0], y) for y in ys[1:])
    name = lib.sr_wino_kernel_name(B, H, W, ci, co, 1, 1).decode()
    print(f"{str((B,ci,H,W,co)):26s} " + "  ".join(f"m{m}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF" for m, t in zip(modes, ts)) +
          f"  ratio {ts[0]/ts[-1]:.3f}  equal={same}  [{name}]", flush=True)
