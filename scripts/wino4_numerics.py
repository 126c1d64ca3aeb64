"""F(4x4, 3x3) / F(2x2, 3x3) Winograd and the direct kernel against an fp64 convolution on input families that stress Winograd
arithmetic: post-ReLU statistics, DC offsets, heavy tails, weight gain, smooth maps.
    python scripts/wino4_numerics.py > profiles/r06_wino4_numerics.txt
Columns: error relative to the output range (the bound tests/test_gpu_wino4.py holds: 2e-5), rms-relative error, and the
largest error relative to the per-output magnitude bound sum |w||x| + |b| + |r| (what a direct fp32 convolution is proportional to)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_wino4 import _run, _ref64
import wino_numerics as WN

dev = "cuda:0"
SHAPES = [(2, 64, 64, 80, 64, True), (1, 192, 48, 64, 64, True), (1, 128, 32, 48, 128, False)]
print("# error of the three 3x3 kernels against an fp64 ATen convolution (conv + bias [+ residual] + LeakyReLU 0.2)")
print("# family        shape                       kernel   range-rel   rms-rel     local-rel")
worst = {}
for shape in SHAPES:
    b, ci, h, w, co, with_res = shape
    for fam in WN.FAMILIES:
        torch.manual_seed(ci + co)
        conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
        with torch.no_grad():
            conv.weight.mul_(WN.weight_gain(fam))
        x = WN.make_input(fam, (b, ci, h, w), dev).contiguous(memory_format=torch.channels_last)
        res = WN.make_input("relu" if "relu" in fam else "randn", (b, co, h, w), dev, seed=7).contiguous(
            memory_format=torch.channels_last) if with_res else None
        with torch.inference_mode():
            ref = _ref64(x, conv, res, 0.2)
            loc = WN.local_scale64(x, conv, res)
            for kind in ("w4_ws", "w2", "direct"):
                y = _run(kind, x, conv, res, 0.2)
                rng, rms, lo = WN.errors(y, ref, loc)
                worst[kind] = max(worst.get(kind, 0.0), rng)
                print(f"{fam:14s} {str(shape):27s} {kind:7s}  {rng:9.2e}  {rms:9.2e}  {lo:9.2e}")
print("# worst range-relative error:", {k: f"{v:.2e}" for k, v in worst.items()})
