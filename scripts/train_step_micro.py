"""Timed WHOLE training step on HIP kernels: DepthModel.forward("train") from images (image-prior + matching encoders in
training mode, metadata-MLP sweep, CVEncoder, DepthDecoderPP, exp) + loss.backward() through every stage.
    python scripts/train_step_micro.py [batch] [views] [freeze_encoders 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simplerecon_amd import depth_model as dm, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
freeze = len(sys.argv) > 3 and sys.argv[3] == "1"
D, H, W = 64, 480, 640
dev = torch.device("cuda", 0)
opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=D)
model = dm.DepthModel(opts)
synthetic.seeded_fill_(model.encoder, seed=6, gain=1.0)
for i, m in enumerate((model.matching_model, model.cost_volume_net, model.depth_decoder, model.cost_volume.mlp)):
    synthetic.seeded_fill_(m, seed=40 + i)
model = model.to(dev).train()
model.freeze_encoders = freeze
inp = synthetic.cost_volume_inputs(B, K, 16, H // 4, W // 4, seed=6, device=dev)
rng = np.random.default_rng(1)
eye = torch.eye(4, device=dev).expand(B, 4, 4).contiguous()
cur = {"image_b3hw": torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype("float32")).to(dev),
       "invK_s1_b44": inp["cur_invK"], "cam_T_world_b44": eye, "world_T_cam_b44": eye}
src = {"image_b3hw": torch.from_numpy(rng.standard_normal((B, K, 3, H, W)).astype("float32")).to(dev),
       "K_s1_b44": inp["src_Ks"], "cam_T_world_b44": inp["src_extrinsics"], "world_T_cam_b44": inp["src_poses"]}


AMP = os.environ.get("SR_TRAIN_AUTOCAST", "")   # "" (fp32), "fp16" or "bf16": the step inside torch.autocast
AMP_DT = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(AMP)


def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=AMP_DT or torch.float16, enabled=AMP_DT is not None):
        out = model("train", cur, src)
        loss = sum(out[f"log_depth_pred_s{i}_b1hw"].float().abs().mean() for i in range(4))
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    step()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
print(f"batch {B}, {K} views, 64 planes, 640x480, encoders {'frozen' if freeze else 'trained (BatchNorm in training mode)'}: "
      f"training step (forward + backward of the whole DepthModel) {t * 1e3:.1f} ms, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
