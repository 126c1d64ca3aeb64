"""Pointwise GEMM (sr_pw_kernel) under every forced launch plan on the hot shapes of the hero step (HIP events)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops, _lib
dev = "cuda:0"
shapes = [(8, 1536, 15, 20, 256, True, True), (8, 960, 30, 40, 160, True, True), (8, 256, 15, 20, 1536, False, False),
          (8, 160, 30, 40, 960, False, False), (8, 192, 240, 320, 64, False, False), (8, 128, 60, 80, 64, False, False),
          (8, 512, 30, 40, 128, True, True), (8, 128, 30, 40, 512, False, False), (8, 192, 120, 160, 48, False, True)]
if os.environ.get("SR_SWEEP_B1", "0") == "1":   # the same layers at batch 1 (the reference's published operating point)
    shapes = [(1,) + s[1:] for s in shapes] + [(1, 256, 30, 40, 128, False, False), (1, 384, 15, 20, 256, False, False),
                                               (1, 256, 60, 80, 64, False, True), (1, 768, 30, 40, 160, True, False)]
TILED = os.environ.get("SR_SWEEP_TILED", "0") == "1"
plans = [None, (1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 4), (4, 1), (5, 1)]
if TILED:
    ops.PW_TILED = "1"
    plans = [None] + [(c, k) for c in range(4) for k in (1, 2, 4, 8)]
else:
    ops.PW_TILED = "0"
lib = _lib.lib()
print("plan:      " + "  ".join(f"{'auto' if p is None else str(p[0]) + 'x' + str(p[1]):>7s}" for p in plans))
for (B, ci, H, W, co, gate, res) in shapes:
    conv = torch.nn.Conv2d(ci, co, 1).to(dev)
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last) if res else None
    g = torch.rand(B, ci, device=dev) if gate else None
    out = ops.empty_nhwc(B, co, H, W, dev)
    row = []
    for pl in plans:
        for k, dflt in (("SR_PW_NT", 0), ("SR_PW_KS", 0), ("SR_PT_CFG", -1), ("SR_PT_KS", 0)):
            _lib.set_option(k, dflt)
        ops._SHAPE_QUERIES.clear()
        if pl is not None:
            nt, ks = C.c_int(0), C.c_int(0)
            if TILED:
                _lib.set_option("SR_PT_CFG", pl[0]); _lib.set_option("SR_PT_KS", pl[1])
                lib.sr_pw_conv_tiled_plan(B * H * W, ci, co, 1, C.byref(nt), C.byref(ks))
            else:
                _lib.set_option("SR_PW_NT", pl[0]); _lib.set_option("SR_PW_KS", pl[1])
                lib.sr_pw_conv_plan(B, H * W, ci, co, C.byref(nt), C.byref(ks))
            if (nt.value, ks.value) != pl:
                row.append("      -"); continue
        with torch.inference_mode():
            f = lambda: ops.conv2d(x, conv, residual=r, gate=g, out=out)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); e1.synchronize()
            row.append(f"{e0.elapsed_time(e1) / 20 * 1e3:7.1f}")
    print(f"{str((B, ci, H, W, co)):28s}" + "  ".join(row), flush=True)
