"""TSDF fusion kernel timing: python scripts/tsdf_micro.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd.tsdf import OurFuser
DEV = "cuda:0"
for label, bounds, B in (("default +-10 m cube (500^3)", None, 8), ("default cube, 1 frame", None, 1),
                         ("room 8x8x3.2 m (200x200x80)", dict(xmin=-4, xmax=4, ymin=-4, ymax=4, zmin=-0.2, zmax=3.0), 8)):
    fuser = OurFuser(bounds=bounds, max_fusion_depth=3.0, device=DEV)
    f = fuser.tsdf_fuser_pred
    g = torch.Generator(device="cpu").manual_seed(0)
    depth = (1.0 + 1.5 * torch.rand((B, 1, 480, 640), generator=g)).to(DEV).half()
    K = torch.eye(4).repeat(B, 1, 1); K[:, 0, 0] = K[:, 1, 1] = 577.87; K[:, 0, 2], K[:, 1, 2] = 320.0, 240.0
    T = torch.eye(4).repeat(B, 1, 1)
    for i in range(B):
        T[i, 0, 3] = 0.1 * i
    K, T = K.to(DEV).half(), T.to(DEV).half()
    for _ in range(2):
        f.integrate_depth(depth, T, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        f.integrate_depth(depth, T, K)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / n
    vox = f.tsdf_values.numel()
    touched = int((f.tsdf_weights > 0).sum())
    nbytes = 4 * vox + 4 * touched + depth.numel() * 2
    print(f"{label}: {vox/1e6:.1f} M voxels, {B} frames, touched {touched/1e6:.2f} M: {t*1e3:.3f} ms/call, "
          f"{nbytes/t/1e9:.0f} GB/s algorithmic, {B/t:.0f} frames/s")
