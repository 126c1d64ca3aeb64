"""Times the cost-volume backward kernels on the GPU (not run in round 1: GPU budget exhausted after verification).

    python scripts/bwd_micro.py [B] [K] [D] [h] [w]        # defaults: 1 7 64 120 160 (cfg2 / cfg3 per frame)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from simplerecon_amd import synthetic
from simplerecon_amd.cost_volume import CostVolumeManager, FeatureVolumeManager

B, K, D, h, w = ([int(a) for a in sys.argv[1:6]] + [1, 7, 64, 120, 160][len(sys.argv) - 1:])[:5]
dev = "cuda:0"
inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=1, device=dev)
R = torch.randn(B, D, h, w, device=dev)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, mgr in (("dot", CostVolumeManager(h, w, num_depth_bins=D)),
                  ("hero", FeatureVolumeManager(h, w, num_depth_bins=D, num_source_views=K))):
    mgr = mgr.to(dev)
    mgr.differentiable = True
    if name == "hero":
        synthetic.seeded_fill_(mgr.mlp, seed=3)
    cur = inp["cur_feats"].clone().requires_grad_()
    src = inp["src_feats"].clone().requires_grad_()
    args = dict(inp, cur_feats=cur, src_feats=src)

    def fwd():
        with torch.no_grad():
            mgr(**inp)

    def fwd_bwd():
        cur.grad = src.grad = None
        (mgr(**args)[0] * R).sum().backward()
    tf, tb = timed(fwd), timed(fwd_bwd)
    print(f"{name:5s} B={B} K={K} D={D} {h}x{w}: forward {tf:8.3f} ms, forward+backward {tb:8.3f} ms "
          f"(backward ~ {tb - tf:8.3f} ms)")
