import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench_workloads as bw
B = int(sys.argv[1])
wl = bw.HeroCfg3(torch.device("cuda", 0), 0, B=B, graph=True, name=f"hero_b{B}_graph")
for i in range(4): wl.step(i)
torch.cuda.synchronize(); t = time.perf_counter()
n = 20
for i in range(n): wl.step(i)
torch.cuda.synchronize(); print(f"B={B} reserve_max_batch={os.environ.get('SR_SWEEP_RESERVE_MAX_POINTS')} {(time.perf_counter()-t)/n*1e3:.3f} ms per step, {(time.perf_counter()-t)/n*1e3/B:.3f} per frame")
