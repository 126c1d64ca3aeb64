"""Split-precision MLP sweep (SR_MLP_SPLIT=bf16|f16, fenced experiment) next to the fp32-MFMA kernel: error of each
against the fp64 oracle on one seeded case, then the sweep time at the hero_cfg3 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import bench_workloads as bw
import oracle
from simplerecon_amd import _lib, synthetic
from simplerecon_amd.cost_volume import FeatureVolumeManager

DEV = "cuda:0"
MODES = [("fp32", None), ("bf16", "bf16"), ("f16", "f16")]


def set_mode(m):
    _lib.set_option("SR_MLP_SPLIT", 0 if m is None else m)


def main():
    B, K, C, h, w, D = 1, 7, 16, int(os.environ.get("SR_CHK_H", 60)), int(os.environ.get("SR_CHK_W", 80)), int(os.environ.get("SR_CHK_D", 32))
    inp = synthetic.cost_volume_inputs(B, K, C, h, w, seed=11)
    mgr = FeatureVolumeManager(h, w, num_depth_bins=D, matching_dim_size=C, num_source_views=K)
    synthetic.seeded_fill_(mgr.mlp, seed=3)
    mgr = mgr.to(DEV)
    sd = {k: v.cpu().numpy() for k, v in mgr.mlp.state_dict().items()}
    mlp = dict(W1=sd["net.0.weight"], b1=sd["net.0.bias"], W2=sd["net.2.weight"], b2=sd["net.2.bias"],
               W3=sd["net.4.weight"], b3=sd["net.4.bias"])
    dinp = {k: v.to(DEV) for k, v in inp.items()}
    outs = {}
    for name, m in MODES:
        set_mode(m)
        with torch.inference_mode():
            vol, lowest, planes, mask = mgr(return_mask=True, **dinp)
        torch.cuda.synchronize()
        outs[name] = (vol.cpu().numpy().astype(np.float64), mask.cpu().numpy())
    set_mode(None)
    n = {k: v.numpy() for k, v in inp.items()}
    planes_np = planes[:, :, 0, 0].cpu().numpy()
    cv64, _, mask64 = oracle.mlp_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["src_poses"],
                                        n["cur_invK"], planes_np, mlp, want_mask=True, precision="f64")
    cv32, _, _ = oracle.mlp_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["src_poses"],
                                   n["cur_invK"], planes_np, mlp, want_mask=True, precision="f32")
    rng = np.abs(cv64).max()
    print(f"case B={B} K={K} {h}x{w} D={D}: |cv| max {rng:.3f}")
    rows = [("oracle f32 (CPU)", cv32.astype(np.float64))] + [(f"HIP {k}", v[0]) for k, v in outs.items()]
    for name, v in rows:
        e = np.abs(v - cv64)
        print(f"  {name:18s} vs f64 oracle: max {e.max()/rng:.3e}  p99 {np.percentile(e, 99)/rng:.3e}  rms {np.sqrt((e**2).mean())/rng:.3e}  (range-relative)")
    for k in ("bf16", "f16"):
        e = np.abs(outs[k][0] - outs["fp32"][0])
        print(f"  HIP {k} vs HIP fp32: max {e.max()/rng:.3e}; mask equal {bool((outs[k][1] == outs['fp32'][1]).all())}")
    # time at the benchmarked shape
    for Bt in (8, 1):
        wl = bw.HeroCfg3(torch.device("cuda", 0), 0, B=Bt)
        for name, m in MODES:
            set_mode(m)
            with torch.inference_mode():
                t = wl._mlp_sweep_time(5)
            print(f"mlp sweep B={Bt} {name}: {t*1e3:.3f} ms")
        set_mode(None)


main()
