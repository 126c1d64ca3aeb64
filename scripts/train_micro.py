"""Timed training-shaped step of the hot path on HIP kernels: hero cost volume (differentiable metadata-MLP sweep) ->
CVEncoder -> DepthDecoderPP -> exp, loss.backward(); frozen encoders' outputs are synthetic inputs.
    python scripts/train_micro.py [batch] [views] """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import depth_model as dm, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
D, h, w = 64, 120, 160
dev = torch.device("cuda", 0)
opts = dm.default_options(image_width=4 * w, image_height=4 * h, model_num_views=K + 1, matching_num_depth_bins=D)
model = dm.DepthModel(opts)
for i, m in enumerate((model.cost_volume_net, model.depth_decoder, model.cost_volume.mlp)):
    synthetic.seeded_fill_(m, seed=40 + i)
model = model.to(dev)
inp = {k: v.to(dev) for k, v in synthetic.cost_volume_inputs(B, K, 16, h, w, seed=6).items()}
pyr = [f.to(dev).contiguous(memory_format=torch.channels_last) for f in synthetic.image_prior_pyramid(B, h, w, seed=6)]
params = list(model.cost_volume_net.parameters()) + list(model.depth_decoder.parameters())
for mlp_grad in (False, True):
    model.cost_volume.differentiable = mlp_grad and K <= 9   # first-version MLP-sweep backward: Cin <= 256
    cur_f, src_f = inp["cur_feats"].clone().requires_grad_(mlp_grad), inp["src_feats"].clone().requires_grad_(mlp_grad)

    def step():
        for p in params:
            p.grad = None
        out = model.hot_path(pyr, cur_f, src_f, inp["src_extrinsics"], inp["src_poses"], inp["src_Ks"], inp["cur_invK"])
        loss = sum(out[f"log_depth_pred_s{i}_b1hw"].abs().mean() for i in range(4)) + out["depth_pred_s0_b1hw"].mean()
        loss.backward()
    with torch.no_grad():
        fwd_only = lambda: model.hot_path(pyr, cur_f.detach(), src_f.detach(), inp["src_extrinsics"], inp["src_poses"],
                                          inp["src_Ks"], inp["cur_invK"])
        for _ in range(2):
            fwd_only()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            fwd_only()
        torch.cuda.synchronize(); t_inf = (time.perf_counter() - t0) / 5
    for _ in range(2):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print(f"batch {B}, {K} views, 64 planes, 640x480: inference hot path {t_inf * 1e3:.1f} ms; training step (forward + "
          f"backward of CVEncoder + DepthDecoderPP{' + MLP sweep' if model.cost_volume.differentiable else ''}) "
          f"{t * 1e3:.1f} ms", flush=True)
