"""F(2x2) (the product's plan per shape, through ops.conv2d with SR_CONV_WINO4=0) vs the F(4x4) forms on every 3x3 shape of the
hero conv stack at batch 8 and 1: the table sr_conv_prefers_wino4's rule is fitted on (profiles/r05_wino4_shape_sweep.txt)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_wino4 import _run
from simplerecon_amd import _lib, ops
from wino_plan_sweep import SHAPES
dev = "cuda:0"
lib = _lib.lib()


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in (8, 1, 64):
    for (ci, H, W, co) in SHAPES:
        if B == 64 and (H, W) != (120, 160):
            continue
        conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
        x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        res = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        out = ops.empty_nhwc(B, co, H, W, dev)
        with torch.inference_mode():
            ops.WINO4_MODE = 0
            ops._SHAPE_QUERIES.clear()
            t2 = timed(lambda: ops.conv2d(x, conv, residual=res, leaky=0.2, out=out))
            t4 = timed(lambda: _run("w4", x, conv, res, 0.2, out=out))
            t4w = timed(lambda: _run("w4_ws", x, conv, res, 0.2, out=out))
        rule = lib.sr_conv_prefers_wino4(B, H, W, ci, co, 1)
        best = min(t4, t4w)
        print(f"B={B:2d} {str((ci, H, W, co)):22s} F(2x2) {t2:7.1f} us   F(4x4) 4-wave {t4:7.1f}  wave-spec {t4w:7.1f}   "
              f"best/F(2x2) {best / t2:5.2f}  rule={rule}", flush=True)
