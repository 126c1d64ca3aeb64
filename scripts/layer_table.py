"""Per-shape timing table of the conv stack (HIP events around every launch)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_workloads as bw
from simplerecon_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ALL = len(sys.argv) > 2 and sys.argv[2] == "all"   # include both encoders (the whole DepthModel.forward)
wl = bw.HeroCfg3(torch.device("cuda", 0), 0, B=B, with_encoder=ALL)
inp = wl.inp
with torch.inference_mode():
    vol = wl.model.cost_volume(cur_feats=inp["cur_feats"], src_feats=inp["src_feats"], src_extrinsics=inp["src_extrinsics"],
                               src_poses=inp["src_poses"], src_Ks=inp["src_Ks"], cur_invK=inp["cur_invK"],
                               min_depth=inp["min_depth"], max_depth=inp["max_depth"])[0]
    for it in range(4):
        ops.PROFILE = [] if it == 3 else None
        pyramid = list(wl.model.encoder(wl.cur_image)) if ALL else wl.pyramid
        if ALL:
            wl.model.compute_matching_feats(wl.cur_image, wl.src_image, False)
        feats = wl.model.cost_volume_net(vol, pyramid[1:])
        wl.model.depth_decoder(pyramid[:1] + feats)
    torch.cuda.synchronize()
rec = ops.PROFILE; ops.PROFILE = None
agg = collections.OrderedDict()
for name, flops, e0, e1, shape, _ex in rec:
    a = agg.setdefault((name, shape), [0, 0.0, 0.0]); a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3
tot = sum(a[2] for a in agg.values())
print(f"B={B} total conv time {tot*1e3:.2f} ms, {sum(a[1] for a in agg.values())/tot/1e12:.1f} TF")
print(f"{'kernel':26s} {'(B,Ci,H,W,Co,k,s)':34s} calls  us/call   TF    % time")
for (name, shape), (c, f, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{name:26s} {str(shape):34s} {c:4d} {t/c*1e6:9.1f} {f/t/1e12:6.1f} {100*t/tot:6.1f}")
