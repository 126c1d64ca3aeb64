"""Winograd layer shapes of the hero conv stack under run-time switches (HIP events): SR_AB_VAR=NAME, SR_AB_VALUES=a,b
times every shape with env NAME=a, NAME=b (switches that are read per call); the library itself is chosen with
SR_HIP_LIBRARY (scripts/build_alt.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops, _lib
dev = "cuda:0"
shapes = [(8, 64, 240, 320, 64), (8, 192, 240, 320, 64), (8, 128, 240, 320, 64), (8, 64, 120, 160, 64), (8, 192, 120, 160, 64),
          (8, 128, 60, 80, 128), (8, 256, 30, 40, 256), (8, 384, 15, 20, 384), (1, 64, 240, 320, 64), (64, 64, 120, 160, 64)]
if os.environ.get("SR_MICRO_SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SR_MICRO_SHAPES"].split(",")]
var = os.environ.get("SR_AB_VAR", "SR_WINO_XCD")
vals = os.environ.get("SR_AB_VALUES", "0,1").split(",")
lib = _lib.lib()
print("library:", _lib.LIB_PATH, " switch:", var, vals, flush=True)
for (B, ci, H, W, co) in shapes:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.empty_nhwc(B, co, H, W, dev)
    ts, ys = [], []
    for v in vals:
        _lib.set_option(var, int(v))   # (the library's option table: include/simplerecon_hip.h)
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
    print(f"{str((B,ci,H,W,co)):26s} " + "  ".join(f"{v}: {t*1e6:8.1f} us {fl/t/1e12*16/36/157.3:5.3f} util" for v, t in zip(vals, ts)) +
          f"  equal={same}", flush=True)
