"""rocBLAS / hipBLASLt fp32 GEMM rates (through torch.mm) for the 1x1-conv shapes of the image-prior encoder, next to
the HIP 1x1 conv kernel on the same shapes: is a library GEMM the better tool for these dense [B*H*W, Cin] x [Cin, Cout]
products?  (BASELINE north star: 'rocBLAS/MFMA only where it is a dense im2col GEMM'.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops
dev = "cuda:0"
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(8 * 30 * 40, 128, 512), (8 * 30 * 40, 160, 960), (8 * 30 * 40, 960, 160), (8 * 15 * 20, 256, 1536),
          (8 * 15 * 20, 1536, 256), (8 * 60 * 80, 64, 256), (8 * 60 * 80, 256, 64), (8 * 240 * 320, 192, 64)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


for M, K, N in shapes:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    t_mm = timed(lambda: torch.mm(a, w.t(), out=out))
    t_addmm = timed(lambda: torch.addmm(bias, a, w.t(), out=out))
    conv = torch.nn.Conv2d(K, N, 1).to(dev)
    x = a.view(1, M // 32, 32, K).permute(0, 3, 1, 2)      # channels-last [1, K, M/32, 32]
    with torch.inference_mode():
        t_hip = timed(lambda: ops.conv2d(x, conv, act="silu"))
    fl = 2.0 * M * K * N
    print(f"M={M:7d} K={K:5d} N={N:5d}: torch.mm {t_mm*1e6:8.1f} us {fl/t_mm/1e12:6.1f} TF | addmm {t_addmm*1e6:8.1f} us | "
          f"HIP 1x1 conv+bias+SiLU {t_hip*1e6:8.1f} us {fl/t_hip/1e12:6.1f} TF", flush=True)
