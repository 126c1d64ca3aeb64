"""sr_mbconv_expand_dw_se_fwd on the three MBConv shapes of EfficientNetV2-S at 640x480, batch 8 (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from simplerecon_amd import image_encoder, ops, synthetic
dev = "cuda:0"
for (b, ci, h, w, mid) in ((8, 128, 30, 40, 512), (8, 160, 30, 40, 960), (8, 256, 15, 20, 1536)):
    pw = synthetic.seeded_fill_(nn.Conv2d(ci, mid, 1, bias=False), seed=1).to(dev)
    bn1 = nn.BatchNorm2d(mid).eval().to(dev)
    dw = synthetic.seeded_fill_(nn.Conv2d(mid, mid, 3, padding=1, groups=mid, bias=False), seed=2).to(dev)
    bn2 = nn.BatchNorm2d(mid).eval().to(dev)
    se = image_encoder.SqueezeExcite(mid, ci // 4).to(dev)
    x = torch.randn(b, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.inference_mode():
        f = lambda: ops.mbconv_expand_dw_se(x, pw, bn1, dw, bn2, se)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
    print(f"{(b, ci, h, w, mid)}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us", end="   ")
print()
