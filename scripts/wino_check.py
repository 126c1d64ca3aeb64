"""Debug helper: Winograd conv vs torch conv2d on the GPU for a few channel counts; prints where they differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from simplerecon_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for (B, ci, H, W, co) in [(1, 16, 16, 32, 64), (1, 32, 16, 32, 64), (1, 48, 16, 32, 64), (1, 64, 16, 32, 64), (2, 64, 24, 48, 64)]:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev)
    with torch.inference_mode():
        y = ops.conv2d(x, conv)
        ref = F.conv2d(x, conv.weight, conv.bias, padding=1)
    err = (y - ref).abs()
    print((B, ci, H, W, co), "max err", float(err.max()), "rel", float(err.max() / ref.abs().max()))
    if err.max() > 1e-3:
        bad = (err > 1e-3)
        print("  bad fraction", float(bad.float().mean()), "per-channel bad:", bad.float().mean(dim=(0, 2, 3))[:8].tolist())
        print("  per-row bad:", [round(v, 2) for v in bad.float().mean(dim=(0, 1, 3)).tolist()])
        print("  per-col bad:", [round(v, 2) for v in bad.float().mean(dim=(0, 1, 2)).tolist()])
