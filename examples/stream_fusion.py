"""End-to-end streaming example on synthetic data: posed frames -> online keyframe / source selection
(simplerecon_amd.keyframes) -> DepthModel.forward (image-prior + matching encoders, plane-sweep cost volume, cost-volume
encoder and UNet++ decoder, all on HIP kernels) -> TSDF fusion of the predicted depth (simplerecon_amd.tsdf).  It mirrors
what the reference's test.py does per scan (test.py:210-410) without datasets, checkpoints or mesh export.

    python examples/stream_fusion.py [--frames 120] [--height 192] [--width 256]

Weights are random, so the depth maps are meaningless -- the point is the data flow and the API.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplerecon_amd import depth_model as dm  # noqa: E402
from simplerecon_amd import keyframes as kf  # noqa: E402
from simplerecon_amd import synthetic  # noqa: E402
from simplerecon_amd.tsdf import OurFuser  # noqa: E402


def intrinsics(width, height, scale):
    """ScanNet-like pinhole intrinsics at 1 / 2**scale of the image resolution."""
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 577.87 * width / 640.0
    K[0, 2], K[1, 2] = width / 2.0, height / 2.0
    K[:2] /= 2 ** scale
    return K


def camera_path(n, seed=0):
    """Slow forward / sideways motion with a little rotation: camera-to-world poses."""
    rng = np.random.default_rng(seed)
    T, out = np.eye(4), []
    for _ in range(n):
        a = 0.01 + 0.004 * rng.standard_normal()
        step = np.eye(4)
        step[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        step[:3, 3] = [0.02, 0.002 * rng.standard_normal(), 0.015]
        T = T @ step
        out.append(T.copy())
    return out


def run(frames=120, height=192, width=256, views=8, device="cuda:0", verbose=True):
    opts = dm.default_options(image_width=width, image_height=height, model_num_views=views)
    model = dm.DepthModel(opts)
    for i, m in enumerate((model.encoder, model.matching_model, model.cost_volume_net, model.depth_decoder,
                           model.cost_volume.mlp)):
        synthetic.seeded_fill_(m, seed=20 + i)
    model = model.to(device).eval()
    fuser = OurFuser(bounds=dict(xmin=-2.0, xmax=6.0, ymin=-2.0, ymax=2.0, zmin=-1.0, zmax=7.0), max_fusion_depth=3.0,
                     device=device)
    cfg = kf.DVMVS_Config
    buf = kf.KeyframeBuffer(cfg.test_keyframe_buffer_size, cfg.test_keyframe_pose_distance, cfg.test_optimal_t_measure,
                            cfg.test_optimal_R_measure, store_return_indices=True)
    K1 = intrinsics(width, height, 1)          # matching resolution = image / 4 -> "s1" of the reference's pyramid
    K_depth = intrinsics(width, height, 1)     # the s0 prediction comes out at image / 2
    invK1 = torch.linalg.inv(K1)
    g = torch.Generator(device="cpu").manual_seed(0)
    predicted = 0
    for i, world_T_cam in enumerate(camera_path(frames)):
        image = torch.randn((3, height, width), generator=g)
        if buf.try_new_keyframe(world_T_cam, image, index=i) != kf.KeyframeBuffer.ADDED:
            continue
        sources = buf.get_best_measurement_frames(views - 1)
        if len(sources) < views - 1:
            continue                              # the reference only predicts with a full tuple
        order = kf.sort_sources_by_pose_penalty(np.linalg.inv(world_T_cam).astype(np.float32),
                                                np.stack([s[0] for s in sources]).astype(np.float32))
        sources = [sources[j] for j in order]
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        cur = {"image_b3hw": image[None].to(device), "invK_s1_b44": invK1[None].to(device),
               "cam_T_world_b44": f32(np.linalg.inv(world_T_cam))[None].to(device),
               "world_T_cam_b44": f32(world_T_cam)[None].to(device)}
        src = {"image_b3hw": torch.stack([s[1] for s in sources])[None].to(device),
               "K_s1_b44": K1[None, None].repeat(1, views - 1, 1, 1).to(device),
               "cam_T_world_b44": f32(np.stack([np.linalg.inv(s[0]) for s in sources]))[None].to(device),
               "world_T_cam_b44": f32(np.stack([s[0] for s in sources]))[None].to(device)}
        with torch.inference_mode():
            out = model("test", cur, src, unbatched_matching_encoder_forward=False, return_mask=True)
            depth = out["depth_pred_s0_b1hw"]
            fuser.fuse_frames(depth, K_depth[None].to(device), cur["cam_T_world_b44"], None)
        predicted += 1
    vol = fuser.tsdf_fuser_pred
    touched = int((vol.tsdf_weights > 0).sum())
    if verbose:
        print(f"{frames} frames -> {predicted} keyframes predicted and fused; TSDF {tuple(vol.shape)}: {touched} voxels touched")
    return predicted, touched


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=256)
    a = ap.parse_args()
    run(a.frames, a.height, a.width)
