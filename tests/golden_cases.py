"""Seeded test-case definitions shared by tests/golden/make_golden.py (which runs the
upstream reference on them) and by the parity tests (which run the oracle and the
HIP path on the same bytes).  Inputs are regenerated from (seed, shape); only the
reference's OUTPUTS are stored under tests/golden/."""
import os

import numpy as np
import torch

from simplerecon_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

VOLUME_CASES = {
    # BASELINE.json configs[0]: dot model, B=1, 2 source views, 32 planes, 320x256 -> 64x80
    "dot_cfg1": dict(model="dot", B=1, K=2, C=16, D=32, h=64, w=80, seed=2),
    "dot_small": dict(model="dot", B=2, K=3, C=16, D=8, h=24, w=32, seed=1),
    "dot_edge": dict(model="dot", B=1, K=4, C=16, D=6, h=20, w=28, seed=5, edge=True),
    "dot_planes": dict(model="dot", B=2, K=2, C=16, D=5, h=16, w=24, seed=6, pixel_planes=True),
    "hero_small": dict(model="hero", B=2, K=3, C=16, D=8, h=24, w=32, seed=3),
    # K=7, C=16 -> the 202-channel MLP of hero_model.yaml
    "hero_k7": dict(model="hero", B=1, K=7, C=16, D=4, h=24, w=32, seed=4),
    "hero_edge": dict(model="hero", B=1, K=4, C=16, D=6, h=20, w=28, seed=7, edge=True),
    "hero_planes": dict(model="hero", B=1, K=2, C=16, D=5, h=16, w=24, seed=8, pixel_planes=True),
}


def _edge_poses(B, K):
    """Poses that hit the reference's special cases: identity (R_measure = 0, samples on
    texel centres), a camera looking backwards (z' < 0 -> mask 0), a large rotation
    (mostly out-of-bounds samples), a pure forward translation."""
    def rot_y(t):
        c, s = np.cos(t), np.sin(t)
        R = np.eye(4)
        R[0, 0], R[0, 2], R[2, 0], R[2, 2] = c, s, -s, c
        return R
    Ts = [np.eye(4), rot_y(np.pi), rot_y(0.6), np.eye(4)]
    Ts[1][:3, 3] = [0.05, 0.0, 0.1]
    Ts[2][:3, 3] = [0.3, 0.1, 0.0]
    Ts[3][:3, 3] = [0.0, 0.0, 0.2]
    src_poses = np.stack([np.stack(Ts[:K])] * B).astype(np.float64)
    return src_poses.astype(np.float32), np.linalg.inv(src_poses).astype(np.float32)


def volume_inputs(case, device="cpu"):
    inp = synthetic.cost_volume_inputs(case["B"], case["K"], case["C"], case["h"], case["w"],
                                       seed=case["seed"], device="cpu")
    if case.get("edge"):
        p, e = _edge_poses(case["B"], case["K"])
        inp["src_poses"], inp["src_extrinsics"] = torch.from_numpy(p), torch.from_numpy(e)
    if case.get("pixel_planes"):
        rng = np.random.default_rng(900 + case["seed"])
        B, D, h, w = case["B"], case["D"], case["h"], case["w"]
        base = np.exp(np.log(0.25) + np.log(5.0 / 0.25) * np.linspace(0, 1, D)).reshape(1, D, 1, 1)
        planes = base * (1.0 + 0.1 * rng.uniform(-1, 1, size=(B, 1, h, w)))
        inp["depth_planes_bdhw"] = torch.from_numpy(planes.astype(np.float32))
    return {k: v.to(device) for k, v in inp.items()}


# gradients of the cost-volume managers (reference autograd; groundwork for the backward pass)
GRAD_CASES = {
    "dot": dict(model="dot", B=1, K=2, C=16, D=4, h=12, w=16, seed=51),
    "hero": dict(model="hero", B=1, K=2, C=16, D=3, h=10, w=12, seed=52),
    # K = 7: the 202-channel MLP of hero_model.yaml (channel layout of the backward chain)
    "hero_k7": dict(model="hero", B=1, K=7, C=16, D=2, h=6, w=8, seed=53),
}


def grad_cotangent(case):
    """dL/d cost_volume of the gradient goldens: L = sum(cost_volume * R), R ~ N(0, 1) seeded."""
    rng = np.random.default_rng(7000 + case["seed"])
    return rng.standard_normal((case["B"], case["D"], case["h"], case["w"]), dtype=np.float32)


def load_golden(prefix, name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{prefix}_{name}.npz")))


# ------------------------------------------------------------------ conv stack

BLOCK_CASES = {
    "same": dict(cin=16, cout=16, stride=1, B=2, h=12, w=20, seed=11),          # identity skip
    "widen": dict(cin=24, cout=16, stride=1, B=1, h=10, w=14, seed=12),         # conv1x1 skip (layers.py:62)
    "down": dict(cin=16, cout=32, stride=2, B=2, h=12, w=20, seed=13),          # conv3x3-s2 skip
    "down_odd": dict(cin=8, cout=8, stride=2, B=1, h=15, w=21, seed=14),        # odd sizes, stride 2
    "wide_in": dict(cin=192, cout=64, stride=1, B=1, h=8, w=12, seed=15),       # decoder in_conv shape
}


def block_input(case, device="cpu"):
    rng = np.random.default_rng(case["seed"])
    x = rng.standard_normal((case["B"], case["cin"], case["h"], case["w"]), dtype=np.float32)
    return torch.from_numpy(x).to(device)


GRAD_BLOCK_CASES = ("same", "widen", "down", "down_odd")   # BasicBlock backward goldens (reference autograd)


def block_cotangent(case, shape):
    rng = np.random.default_rng(8000 + case["seed"])
    return rng.standard_normal(shape, dtype=np.float32)


def upsample_input(device="cpu"):
    rng = np.random.default_rng(21)
    return torch.from_numpy(rng.standard_normal((2, 5, 7, 9), dtype=np.float32)).to(device)


NET_CASES = {
    # real channel widths of hero_model.yaml (depth_model.py:123-135), tiny spatial size:
    # matching res 16x24 -> pyramid 32x48, 16x24, 8x12, 4x6, 2x3
    "full_width": dict(D=64, enc_ch=[24, 48, 64, 160, 256], cv_outs=[64, 128, 256, 384], B=1, h=16, w=24, seed=31),
    # batch 2, non-square, fewer planes
    "narrow": dict(D=8, enc_ch=[6, 10, 12, 20, 28], cv_outs=[64, 128, 256, 384], B=2, h=8, w=16, seed=33),
}


def net_inputs(case, device="cpu"):
    rng = np.random.default_rng(case["seed"])
    vol = torch.from_numpy(rng.standard_normal((case["B"], case["D"], case["h"], case["w"]), dtype=np.float32))
    feats = synthetic.image_prior_pyramid(case["B"], case["h"], case["w"], chans=case["enc_ch"], seed=case["seed"])
    return vol.to(device), [f.to(device) for f in feats]


# end-to-end training chain (reference train.py:126-145 behind the encoders): FeatureVolumeManager -> CVEncoder ->
# DepthDecoderPP -> exp, loss = sum_k sum(exp(log_depth_k) * R_k); golden = the reference's autograd in float64
CHAIN_CASE = dict(B=1, K=2, C=16, D=8, h=8, w=16, seed=61, enc_ch=[6, 10, 12, 20, 28], cv_outs=[64, 128, 256, 384])
CHAIN_ENC_PARAMS = ("convs.ds_conv_0.conv1.weight", "convs.conv_0.0.downsample.0.weight", "convs.conv_1.1.conv2.bias",
                    "convs.ds_conv_3.conv1.bias")


def chain_cotangents(case, shapes):
    rng = np.random.default_rng(8400 + case["seed"])
    return {k: rng.standard_normal(shapes[k], dtype=np.float32) for k in sorted(shapes)}


DECODER_GRAD_PARAMS = ("convs.output_0.1.weight", "convs.output_3.0.conv1.bias", "convs.in_conv_04.0.conv1.bias",
                       "convs.right_conv_00.conv1.weight", "convs.diag_conv_40.conv2.bias", "convs.up_conv_12.conv1.weight",
                       "convs.in_conv_31.conv_0.conv2.bias")


def decoder_inputs(case):
    """Five feature maps with the decoder's input widths ([enc_ch[0]] + cv_outs) at strides 1, 1, 2, 4, 8 of the
    matching resolution x2 pyramid used by the forward goldens (seeded)."""
    rng = np.random.default_rng(8200 + case["seed"])
    chans = case["enc_ch"][:1] + case["cv_outs"]
    h, w = 2 * case["h"], 2 * case["w"]
    return [torch.from_numpy(rng.standard_normal((case["B"], c, h >> i, w >> i), dtype=np.float32))
            for i, c in enumerate(chans)]


def decoder_cotangents(case, shapes):
    rng = np.random.default_rng(8300 + case["seed"])
    return {k: rng.standard_normal(shapes[k], dtype=np.float32) for k in sorted(shapes)}


def cv_encoder_cotangents(case, shapes):
    rng = np.random.default_rng(8100 + case["seed"])
    return [rng.standard_normal(s, dtype=np.float32) for s in shapes]


# ------------------------------------------------------------ matching encoder (a16)

MATCHING_CASES = {
    # image [B,3,H,W] -> features [B,16,H/4,W/4]
    "small": dict(B=2, H=64, W=96, seed=41),
    "ragged": dict(B=1, H=40, W=72, seed=42),     # H/4 = 10, W/4 = 18: partial tiles in every kernel
}


def matching_input(case, device="cpu"):
    rng = np.random.default_rng(case["seed"])
    x = rng.standard_normal((case["B"], 3, case["H"], case["W"]), dtype=np.float32)
    return torch.from_numpy(x).to(device)


def encoder_cotangent(case, shape):
    """Seeded cotangent of an encoder output (gradient goldens of the training path)."""
    return np.random.default_rng(case["seed"] + 500).standard_normal(shape).astype(np.float32)


# ------------------------------------------------------------ TSDF fusion (§8f "next" #2)

TSDF_CASES = {
    # a room-sized slab of voxels seen by 5 frames (two integrate_depth calls: batch of 3, then 2 with a depth mask)
    "room": dict(bounds=dict(xmin=-1.0, xmax=1.0, ymin=-0.8, ymax=0.8, zmin=0.2, zmax=2.2), voxel_size=0.04,
                 H=48, W=64, frames=5, max_depth=3.0, seed=51),
    # coarse voxels, frames looking partly away from the volume / from inside it (z <= 0 voxels, out-of-image pixels)
    "skew": dict(bounds=dict(xmin=-0.5, xmax=0.9, ymin=-0.7, ymax=0.4, zmin=-0.4, zmax=1.1), voxel_size=0.05,
                 H=30, W=44, frames=4, max_depth=2.0, seed=52),
}


def tsdf_inputs(case):
    """Depth maps [F,1,H,W], intrinsics and extrinsics [F,4,4] (float32; the fuser receives them through .half())
    and a boolean depth mask for the second half of the frames."""
    rng = np.random.default_rng(case["seed"])
    F_, H, W = case["frames"], case["H"], case["W"]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    depth = np.stack([1.1 + 0.25 * np.sin(0.13 * xx + i) * np.cos(0.11 * yy - i) +
                      0.05 * rng.standard_normal((H, W)).astype(np.float32) for i in range(F_)])[:, None]
    depth[:, :, :2, :3] = 0.0                                  # invalid (zero) depths
    K = np.tile(np.eye(4, dtype=np.float32), (F_, 1, 1))
    K[:, 0, 0] = K[:, 1, 1] = 0.95 * W
    K[:, 0, 2], K[:, 1, 2] = W / 2, H / 2
    T = np.tile(np.eye(4, dtype=np.float32), (F_, 1, 1))
    for i in range(F_):
        a, b2 = 0.12 * i - 0.1, 0.07 * i
        Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        Rx = np.array([[1, 0, 0], [0, np.cos(b2), -np.sin(b2)], [0, np.sin(b2), np.cos(b2)]], np.float32)
        T[i, :3, :3] = Ry @ Rx
        T[i, :3, 3] = [0.06 * i - 0.1, 0.03 * i, 0.08 * i - 0.05]
    mask = rng.random((F_, 1, H, W)) > 0.2
    return torch.from_numpy(depth.astype(np.float32)), torch.from_numpy(K), torch.from_numpy(T), torch.from_numpy(mask)


# ------------------------------------------------------------ keyframe selection (§8f "next" #4)

def keyframe_stream(seed=61, n=400):
    """A hand-held style camera trajectory (camera-to-world poses) with pauses, fast segments, tracking drop-outs (NaN
    poses, one longer than the 30-frame 'lost' threshold) and a gap in the valid-frame distances."""
    rng = np.random.default_rng(seed)
    poses, dist = [], []
    T = np.eye(4)
    for i in range(n):
        speed = 0.0 if 120 <= i < 150 else (0.06 if 200 <= i < 230 else 0.012)
        w = rng.standard_normal(3) * 0.01 + np.array([0.0, 0.004, 0.0])
        th = np.linalg.norm(w)
        kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        dR = np.eye(3) + np.sin(th) / th * kx + (1 - np.cos(th)) / th ** 2 * kx @ kx
        step = np.eye(4)
        step[:3, :3] = dR
        step[:3, 3] = speed * (np.array([1.0, 0.1, 0.3]) + 0.3 * rng.standard_normal(3))
        T = T @ step
        pose = T.copy()
        if 60 <= i < 70 or 260 <= i < 300:
            pose[:] = np.nan
        poses.append(pose)
        dist.append(45 if i == 330 else 1)
    return poses, dist
