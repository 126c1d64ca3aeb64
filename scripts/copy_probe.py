import sys, os
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench_workloads
from torch.profiler import profile, ProfilerActivity
wl = bench_workloads.WORKLOADS["hero_cfg3"](torch.device("cuda", 0), 0)
with torch.inference_mode():
    for i in range(3): wl.step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        wl.step(0)
        torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::to", "aten::fill_", "aten::zero_", "aten::empty", "aten::zeros") :
        st = [s for s in (ev.stack or []) if "simplerecon_amd" in s or "bench_workloads" in s]
        c[(ev.name, st[0] if st else "?")] += 1
for k, v in c.most_common(25): print(v, k)
