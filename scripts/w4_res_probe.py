import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from test_gpu_wino4 import _run
from simplerecon_amd import ops
dev = "cuda:0"
for (B, ci, H, W, co) in [(8, 64, 240, 320, 64), (8, 192, 240, 320, 64)]:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.empty_nhwc(B, co, H, W, dev)
    for label, r, o in (("res", res, out), ("nores", None, out), ("res=out(in place: no third tensor)", out, out)):
        with torch.inference_mode():
            f = lambda: _run("w4_ws", x, conv, r, 0.2, out=o)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); e1.synchronize()
            print((B, ci, H, W, co), label, f"{e0.elapsed_time(e1) * 1e3 / 20:.1f} us")
