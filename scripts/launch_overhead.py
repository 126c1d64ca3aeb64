"""CPU submission time vs GPU time of one step (is the step launch-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_workloads as bw
for name in sys.argv[1:] or ["hero_b1_core", "hero_b1", "hero_cfg3"]:
    wl = bw.WORKLOADS[name](torch.device("cuda", 0), 0)
    with torch.inference_mode():
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            wl.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{name}: CPU submit {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step")
