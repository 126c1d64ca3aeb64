import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from simplerecon_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
B, ci, H, W, co = 1, 16, 240, 320, 24
conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
x = torch.randn(B, ci, H, W, device=dev)
with torch.inference_mode():
    ref = F.conv2d(x, conv.weight, conv.bias, padding=1)
    for trial in range(3):
        out = ops.empty_nhwc(B, co, H, W, dev).fill_(777.0)
        ops.conv2d(x, conv, out=out)
        torch.cuda.synchronize()
        bad = (out - ref).abs() > 1e-3
        idx = bad.nonzero()
        print("trial", trial, "bad", int(bad.sum()), "sentinel left:", int((out == 777.0).sum()))
        for t in idx[:12].tolist():
            b, c, y, xx = t
            print("   ", t, "got", float(out[b, c, y, xx]), "want", float(ref[b, c, y, xx]), "region", (y // 8, xx // 16), "tile", ((y % 8) // 2, (xx % 16) // 2), "q", (y % 2) * 2 + xx % 2)
