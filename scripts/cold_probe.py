"""How much of a small layer's in-model time is cache state?  The same launch (HIP events around it) with its input
(a) re-read in a tight loop (what the isolated sweeps measure), (b) freshly written by another kernel just before (what the
model does), (c) evicted from the Infinity Cache by a 1-GB fill in between."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops
dev = "cuda:0"
shapes = [(8, 128, 60, 80, 64, 1), (8, 128, 60, 80, 64, 3), (8, 64, 60, 80, 64, 3), (8, 256, 30, 40, 128, 1), (8, 384, 15, 20, 256, 1),
          (8, 1536, 15, 20, 256, 1), (8, 256, 15, 20, 1536, 1), (8, 960, 30, 40, 160, 1), (8, 128, 60, 80, 128, 3), (8, 256, 30, 40, 128, 3),
          (8, 64, 120, 160, 64, 3), (8, 64, 240, 320, 64, 3)]
big = torch.empty(256 << 20, dtype=torch.float32, device=dev)   # 1 GB
for (B, ci, H, W, co, k) in shapes:
    conv = torch.nn.Conv2d(ci, co, k, padding=k // 2).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    src = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.empty_nhwc(B, co, H, W, dev)
    res = {}
    with torch.inference_mode():
        f = lambda: ops.conv2d(x, conv, leaky=0.2, out=out)
        for _ in range(3): f()
        for mode in ("hot", "fresh", "evicted"):
            ts = []
            for it in range(12):
                if mode == "fresh":
                    x.copy_(src)
                elif mode == "evicted":
                    x.copy_(src); big.fill_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            res[mode] = ts[len(ts) // 2]
    print(f"{str((B,ci,H,W,co,k)):32s} hot {res['hot']:7.1f} us   input just written {res['fresh']:7.1f}   after a 1-GB fill {res['evicted']:7.1f}", flush=True)
