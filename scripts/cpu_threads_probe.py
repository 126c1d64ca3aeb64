"""How many threads should the ATen CPU baseline use on this host?  (one plane of the hero sweep + one BasicBlock)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_cpu_aten as aten
from simplerecon_amd import synthetic
from simplerecon_amd.networks import MLP
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
B, K, h, w = 1, 7, 120, 160
inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=0)
mlp = MLP([202, 128, 128, 1], disable_final_activation=True)
lin = [(m.weight, m.bias) for m in mlp.net if isinstance(m, torch.nn.Linear)]
planes = torch.tensor([[1.0, 2.0]])
x = torch.randn(1, 64, 240, 320); wgt = torch.randn(64, 64, 3, 3)
for nt in (8, 16, 32, 64, 128, 256):
    if nt > os.cpu_count():
        break
    torch.set_num_threads(nt)
    with torch.inference_mode():
        aten.mlp_volume(inp["cur_feats"], inp["src_feats"], inp["src_Ks"], inp["src_extrinsics"], inp["src_poses"], inp["cur_invK"], planes[:, :1], lin)
        t0 = time.perf_counter()
        aten.mlp_volume(inp["cur_feats"], inp["src_feats"], inp["src_Ks"], inp["src_extrinsics"], inp["src_poses"], inp["cur_invK"], planes, lin)
        t1 = time.perf_counter()
        torch.nn.functional.conv2d(x, wgt, padding=1)
        t2 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.conv2d(x, wgt, padding=1)
        t3 = time.perf_counter()
    print(f"threads {nt}: sweep {(t1 - t0) / 2:.3f} s/plane, conv64 240x320 {(t3 - t2) / 3 * 1e3:.1f} ms", flush=True)
