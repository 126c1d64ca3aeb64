"""Per-shape timing table of the pointwise (1x1) GEMM launches of one hero step (HIP events around every launch)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_workloads as bw
from simplerecon_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = bw.HeroCfg3(torch.device("cuda", 0), 0, B=B)
wl.model.prior_on_side_stream = False
with torch.inference_mode():
    for it in range(4):
        ops.PROFILE = [] if it == 3 else None
        wl.step()
    torch.cuda.synchronize()
rec = ops.PROFILE; ops.PROFILE = None
agg = collections.OrderedDict()
for name, flops, e0, e1, shape, ex in rec:
    if "pw" not in name and "GEMM" not in name:
        continue
    a = agg.setdefault((name, shape), [0, 0.0, 0.0]); a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3
tot = sum(a[2] for a in agg.values())
print(f"B={B} total 1x1 time {tot*1e3:.2f} ms, {sum(a[1] for a in agg.values())/tot/1e12:.1f} TF")
print(f"{'kernel':30s} {'(B,Ci,H,W,Co,k,s,Ho,Wo,res)':44s} calls  us/call   TF   GB/s  % time")
for (name, shape), (c, f, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    b, ci, h, w, co = shape[:5]
    byts = 4.0 * b * h * w * (ci + co * (2 if shape[-1] else 1)) * c
    print(f"{name:30s} {str(shape):44s} {c:4d} {t/c*1e6:9.1f} {f/t/1e12:6.1f} {byts/t/1e9:6.0f} {100*t/tot:6.1f}")
