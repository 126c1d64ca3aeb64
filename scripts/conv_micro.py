"""Single-shape conv micro-benchmark (HIP events), for ablations via SR_CONV_DEBUG."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops
dev = "cuda:0"
shapes = [(8, 64, 240, 320, 64, 3), (8, 192, 240, 320, 64, 3), (8, 64, 120, 160, 64, 3), (8, 192, 240, 320, 64, 1),
          (8, 256, 30, 40, 256, 3), (8, 384, 15, 20, 384, 3), (1, 64, 240, 320, 64, 3)]
if os.environ.get("SR_MICRO_SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SR_MICRO_SHAPES"].split(",")]
for (B, ci, H, W, co, k) in shapes:
    conv = torch.nn.Conv2d(ci, co, k, padding=k // 2).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.empty_nhwc(B, co, H, W, dev)
    for with_res in (False, True):
        with torch.inference_mode():
            f = lambda: ops.conv2d(x, conv, residual=res if with_res else None, leaky=0.2, out=out)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n): f()
            e1.record(); e1.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / n
        fl = 2.0 * B * H * W * co * ci * k * k
        print(f"dbg={os.environ.get('SR_CONV_DEBUG','0'):>2s} {str((B,ci,H,W,co,k)):30s} res={int(with_res)} {t*1e6:9.1f} us {fl/t/1e12:7.1f} TF", flush=True)
