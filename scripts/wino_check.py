"""Debug helper: Winograd conv vs torch conv2d on the GPU for a few shapes; prints where they differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from simplerecon_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for (B, ci, H, W, co, act) in [(1, 24, 240, 320, 24, "silu"), (1, 24, 240, 320, 24, None), (1, 24, 240, 320, 64, None),
                               (1, 32, 240, 320, 24, None), (1, 24, 64, 96, 24, None), (2, 24, 240, 320, 24, None),
                               (1, 24, 240, 320, 32, None), (1, 16, 240, 320, 24, None)]:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev)
    with torch.inference_mode():
        y = ops.conv2d(x, conv, act=act)
        ref = F.conv2d(x, conv.weight, conv.bias, padding=1)
        if act == "silu":
            ref = F.silu(ref)
    err = (y - ref).abs()
    print((B, ci, H, W, co, act), "max err", float(err.max()), "rel", float(err.max() / ref.abs().max()))
    if err.max() > 1e-3:
        bad = (err > 1e-3)
        print("  bad fraction", float(bad.float().mean()), "per-channel bad:", [round(v, 2) for v in bad.float().mean(dim=(0, 2, 3)).tolist()])
        rows = bad.float().mean(dim=(0, 1, 3)).tolist(); cols = bad.float().mean(dim=(0, 1, 2)).tolist()
        print("  bad rows:", [i for i, v in enumerate(rows) if v > 0][:40])
        print("  bad cols:", [i for i, v in enumerate(cols) if v > 0][:40])
