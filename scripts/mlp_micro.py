"""MLP-sweep micro-benchmark (HIP events around sr_mlp_volume_sweep)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_workloads as bw
for B in (8, 1):
    wl = bw.HeroCfg3(torch.device("cuda", 0), 0, B=B)
    with torch.inference_mode():
        t = wl._mlp_sweep_time(5)
    N = wl.h * wl.w
    fl = 2.0 * (202 * 128 + 128 * 128 + 128) * B * wl.D * N
    print(f"mlp sweep B={B}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TF (algorithmic)")
for B in (8, 1):
    wl = bw.DotCfg2(torch.device("cuda", 0), 0, B=B)
    with torch.inference_mode():
        r = wl.roofline(20)
    print(f"dot sweep B={B}: {r['avg_launch_us']:.1f} us  gather {r['onchip_gather_GBps']/1e3:.1f} TB/s  hbm-alg {r['achieved']:.1f} GB/s")
