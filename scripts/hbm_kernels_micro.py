"""The HBM-bound helpers of the hero step at their batch-8 / 64-image shapes, old and new form of each (HIP events):
maxblurpool [64,64,240,320] (SR_POOL_STREAM), bilinear x2 upsample into a concat slice (SR_UPSAMPLE_QUAD)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from simplerecon_amd import _lib, ops

dev = "cuda:0"


def t(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


with torch.inference_mode():
    for (b, c, h, w) in ((64, 64, 240, 320), (8, 64, 240, 320)):
        x = ops.empty_nhwc(b, c, h, w, dev).normal_()
        out = ops.empty_nhwc(b, c, h // 2, w // 2, dev)
        gb = (x.numel() + out.numel()) * 4 / 1e9
        row = []
        for v in (0, 1):
            _lib.set_option("SR_POOL_STREAM", v)
            s = t(lambda: ops.maxblurpool(x, out=out))
            row.append(f"stream={v}: {s*1e6:7.1f} us {gb/s/1e3:5.2f} TB/s")
        print(f"maxblurpool {(b, c, h, w)}: " + "   ".join(row), flush=True)
        del x, out
    for (b, c, h, w, ctot) in ((8, 64, 120, 160, 192), (8, 64, 120, 160, 64), (8, 64, 60, 80, 192), (8, 128, 30, 40, 384), (8, 256, 15, 20, 768),
                               (1, 64, 120, 160, 192)):
        x = ops.empty_nhwc(b, c, h, w, dev).normal_()
        buf = ops.empty_nhwc(b, ctot, 2 * h, 2 * w, dev)
        out = buf[:, :c]
        gb = (x.numel() + out.numel()) * 4 / 1e9
        row = []
        for v in (0, 1):
            _lib.set_option("SR_UPSAMPLE_QUAD", v)
            s = t(lambda: ops.upsample2x(x, out=out))
            row.append(f"quad={v}: {s*1e6:7.1f} us {gb/s/1e3:5.2f} TB/s")
        print(f"upsample2x {(b, c, h, w)} into {ctot} channels: " + "   ".join(row), flush=True)
