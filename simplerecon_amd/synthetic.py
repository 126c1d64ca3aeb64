"""Deterministic synthetic inputs for the plane-sweep hot path.

ScanNet-shaped geometry: intrinsics fx = fy = 577.87 * (w / 640) with the
principal point at the image centre (cf. reference datasets/scannet_dataset.py:
448-470, K scaled by the feature-map width), the reference camera at the
identity, source view k rotated 0.05*(k+1) rad about y and translated
(0.10, -0.03, 0.02)*(k+1) m (DVMVS-like baselines, reference
tools/keyframe_buffer.py:12-22), N(0,1) matching features (InstanceNorm output,
reference modules/networks.py:201) and 0.25 m .. 5.0 m matching depths
(reference options.py:133-134).

Everything is generated with numpy so that the same seed gives the same bytes on
the build container, the GPU box and inside the golden-vector generator.
"""
import zlib

import numpy as np
import torch

MIN_DEPTH = 0.25
MAX_DEPTH = 5.0


def intrinsics(h, w, dtype=np.float32):
    """4x4 pinhole K at feature-map resolution h x w (and its inverse)."""
    K = np.eye(4, dtype=np.float64)
    f = 577.87 * (w / 640.0)
    K[0, 0] = f
    K[1, 1] = f
    K[0, 2] = w / 2.0
    K[1, 2] = h / 2.0
    return K.astype(dtype), np.linalg.inv(K).astype(dtype)


def _rot_y(theta):
    c, s = np.cos(theta), np.sin(theta)
    R = np.eye(4)
    R[0, 0], R[0, 2], R[2, 0], R[2, 2] = c, s, -s, c
    return R


def _small_rot(rng, scale):
    """Random small rotation (Rodrigues) for per-batch jitter."""
    v = rng.normal(size=3) * scale
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def poses(B, K, seed=0, jitter=0.02):
    """Returns (src_poses = cur_cam_T_src_cam, src_extrinsics = src_cam_T_cur_cam), [B,K,4,4] fp32.

    Batch element 0 is the exact §8d layout; further elements add a seeded jitter so
    that a batch is not B copies of the same geometry."""
    rng = np.random.default_rng(1000 + seed)
    src_poses = np.zeros((B, K, 4, 4))
    for b in range(B):
        for k in range(K):
            T = _rot_y(0.05 * (k + 1))
            T[:3, 3] = np.array([0.10, -0.03, 0.02]) * (k + 1)
            if b > 0 and jitter > 0:
                J = np.eye(4)
                J[:3, :3] = _small_rot(rng, jitter)
                J[:3, 3] = rng.normal(size=3) * jitter
                T = J @ T
            src_poses[b, k] = T
    src_extr = np.linalg.inv(src_poses)
    return src_poses.astype(np.float32), src_extr.astype(np.float32)


def keyframe_poses(ids, K, jitter=0.02):
    """Poses of the keyframes of a STREAM: a function of the keyframe id alone (so that a keyframe gets the same
    cameras whatever rank / batch it lands in): the §8d layout composed with a small rigid jitter seeded by the id.
    Returns (src_poses, src_extrinsics), [len(ids),K,4,4] fp32."""
    base = np.zeros((K, 4, 4))
    for k in range(K):
        T = _rot_y(0.05 * (k + 1))
        T[:3, 3] = np.array([0.10, -0.03, 0.02]) * (k + 1)
        base[k] = T
    src_poses = np.zeros((len(ids), K, 4, 4))
    for j, i in enumerate(ids):
        rng = np.random.default_rng(500000 + int(i))
        for k in range(K):
            J = np.eye(4)
            J[:3, :3] = _small_rot(rng, jitter)
            J[:3, 3] = rng.normal(size=3) * jitter
            src_poses[j, k] = J @ base[k]
    src_extr = np.linalg.inv(src_poses)
    return src_poses.astype(np.float32), src_extr.astype(np.float32)


def cost_volume_inputs(B, K, C, h, w, seed=0, device="cpu", jitter=0.02):
    """Keyword arguments of CostVolumeManager.forward (reference cost_volume.py:345-357)."""
    rng = np.random.default_rng(seed)
    cur = rng.standard_normal((B, C, h, w), dtype=np.float32)
    src = rng.standard_normal((B, K, C, h, w), dtype=np.float32)
    Kmat, invK = intrinsics(h, w)
    src_poses, src_extr = poses(B, K, seed, jitter)
    out = dict(
        cur_feats=torch.from_numpy(cur),
        src_feats=torch.from_numpy(src),
        src_extrinsics=torch.from_numpy(src_extr),
        src_poses=torch.from_numpy(src_poses),
        src_Ks=torch.from_numpy(np.broadcast_to(Kmat, (B, K, 4, 4)).copy()),
        cur_invK=torch.from_numpy(np.broadcast_to(invK, (B, 4, 4)).copy()),
        min_depth=torch.tensor(MIN_DEPTH, dtype=torch.float32).view(1, 1, 1, 1),
        max_depth=torch.tensor(MAX_DEPTH, dtype=torch.float32).view(1, 1, 1, 1),
    )
    return {k: v.to(device) for k, v in out.items()}


def image_prior_pyramid(B, h, w, chans=(24, 48, 64, 160, 256), seed=0, device="cpu"):
    """Stand-in for the image-prior encoder's 5-scale pyramid (reference depth_model.py:346:
    EfficientNetV2-S features, channels [24,48,64,160,256] at 1/2 .. 1/32 of the image).
    (h, w) is the MATCHING resolution (= image / 4): scales are 2h, h, h/2, h/4, h/8."""
    rng = np.random.default_rng(77 + seed)
    feats = []
    for i, c in enumerate(chans):
        hh, ww = (2 * h) >> i, (2 * w) >> i
        feats.append(torch.from_numpy(rng.standard_normal((B, c, hh, ww), dtype=np.float32)).to(device))
    return feats


def seeded_fill_(module, seed=0, gain=1.0):
    """Deterministically (re)initialises every parameter of `module` from a numpy RNG keyed
    by the parameter NAME, so two modules with the same state-dict layout (ours and the
    reference's) get bit-identical weights without shipping them.  Weights ~ U(-a, a) with
    a = gain*sqrt(3/fan_in) (variance-preserving), biases ~ U(-0.1, 0.1).  BatchNorm layers get
    non-trivial statistics: weight, running_var ~ U(0.5, 1.5); bias, running_mean ~ U(-0.2, 0.2)."""
    bn_scale, bn_shift = set(), set()
    for mname, m in module.named_modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            pre = mname + "." if mname else ""
            bn_scale.update({pre + "weight", pre + "running_var"})
            bn_shift.update({pre + "bias", pre + "running_mean"})
    with torch.no_grad():
        named = list(module.named_parameters())
        named += [(n, b) for n, b in module.named_buffers() if n in bn_scale or n in bn_shift]
        for name, p in named:
            rng = np.random.default_rng((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
            if name in bn_scale:
                v = rng.uniform(0.5, 1.5, size=tuple(p.shape)).astype(np.float32)
            elif name in bn_shift:
                v = rng.uniform(-0.2, 0.2, size=tuple(p.shape)).astype(np.float32)
            elif p.dim() > 1:
                fan_in = int(np.prod(p.shape[1:]))
                a = gain * np.sqrt(3.0 / fan_in)
                v = rng.uniform(-a, a, size=tuple(p.shape)).astype(np.float32)
            else:
                v = rng.uniform(-0.1, 0.1, size=tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(v).to(p.device))
    return module
