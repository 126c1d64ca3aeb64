"""Geometry helpers of the hot path, API-compatible with reference utils/geometry_utils.py.

The modules hold the reference's buffers (`pix_coords_13N`, `eps`: checkpoints load with strict=True).  Inside the plane
sweeps their arithmetic is fused into the HIP kernels (csrc/sr_common.h: sr_project_sample_xy); called on their own they
run the standalone kernels of csrc/sr_geometry.hip, which use the same operation order -- there is no torch / CPU
fallback (device fp32 tensors only, like the rest of the package).  BackprojectDepth is differentiable in the depth map
and Project3D in the points, as the reference's multi-view losses need; cameras are data."""
import ctypes as C

import torch
from torch import nn

from . import _lib


def _f32c(name, t):
    _lib.require_device_f32(name, t)
    _lib.refuse_autograd(t)
    return t.contiguous()


def _data(name, t):
    """Intrinsics / poses: data on this path (the reference does not optimise them either) -- no gradient is produced."""
    _lib.require_device_f32(name, t)
    if torch.is_grad_enabled() and t.requires_grad:
        raise _lib.HipLibraryError(f"{name} requires a gradient: cameras are data for the HIP geometry helpers")
    return t.detach().contiguous()


def _diff(name, t):
    _lib.require_device_f32(name, t)
    return t.contiguous()


class _Backproject(torch.autograd.Function):
    """d cam_points / d depth (what the reference's multi-view losses differentiate); csrc/sr_geometry.hip."""

    @staticmethod
    def forward(ctx, depth, invK, height, width):
        b = depth.shape[0]
        out = torch.empty((b, 4, height * width), dtype=torch.float32, device=depth.device)
        with _lib.on_device(depth.device):
            rc = _lib.lib().sr_backproject_fwd(_lib.ptr(depth), _lib.ptr(invK), _lib.ptr(out), b, height, width,
                                               _lib.stream_ptr(depth.device))
        _lib.check(rc, "sr_backproject_fwd")
        ctx.save_for_backward(invK)
        ctx.hw, ctx.depth_shape = (height, width), tuple(depth.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (invK,) = ctx.saved_tensors
        g = g.contiguous()
        h, w = ctx.hw
        d_depth = torch.empty(ctx.depth_shape, dtype=torch.float32, device=g.device)
        with _lib.on_device(g.device):
            rc = _lib.lib().sr_backproject_bwd(_lib.ptr(g), _lib.ptr(invK), _lib.ptr(d_depth), g.shape[0], h, w,
                                               _lib.stream_ptr(g.device))
        _lib.check(rc, "sr_backproject_bwd")
        return d_depth, None, None, None


class _Project(torch.autograd.Function):
    """d (pixel x, pixel y, depth) / d points; csrc/sr_geometry.hip."""

    @staticmethod
    def forward(ctx, pts, K, T, eps):
        b, _, n = pts.shape
        out = torch.empty((b, 3, n), dtype=torch.float32, device=pts.device)
        with _lib.on_device(pts.device):
            rc = _lib.lib().sr_project3d_fwd(_lib.ptr(pts), _lib.ptr(K), _lib.ptr(T), _lib.ptr(out), b, n, C.c_float(eps),
                                             _lib.stream_ptr(pts.device))
        _lib.check(rc, "sr_project3d_fwd")
        ctx.save_for_backward(pts, K, T)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, g):
        pts, K, T = ctx.saved_tensors
        g = g.contiguous()
        b, _, n = pts.shape
        d_pts = torch.empty_like(pts)
        with _lib.on_device(g.device):
            rc = _lib.lib().sr_project3d_bwd(_lib.ptr(g), _lib.ptr(pts), _lib.ptr(K), _lib.ptr(T), _lib.ptr(d_pts), b, n,
                                             C.c_float(ctx.eps), _lib.stream_ptr(g.device))
        _lib.check(rc, "sr_project3d_bwd")
        return d_pts, None, None, None


class BackprojectDepth(nn.Module):
    """Holds `pix_coords_13N` (pixel centres at +0.5, reference geometry_utils.py:34-48)."""

    def __init__(self, height: int, width: int):
        super().__init__()
        self.height, self.width = height, width
        ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
        pix = torch.stack([xs.flatten() + 0.5, ys.flatten() + 0.5, torch.ones(height * width)], 0)
        self.register_buffer("pix_coords_13N", pix.unsqueeze(0).float())

    def forward(self, depth_b1hw, invK_b44):
        """depth [B,1,h,w], invK [B,4,4] -> homogeneous camera points [B,4,h*w] (reference geometry_utils.py:51-59)."""
        depth, invK = _diff("depth_b1hw", depth_b1hw), _data("invK_b44", invK_b44)
        b = depth.shape[0]
        if tuple(depth.shape[-2:]) != (self.height, self.width) or depth.numel() != b * self.height * self.width:
            raise ValueError(f"depth map {tuple(depth.shape)} does not match {self.height}x{self.width}")
        return _Backproject.apply(depth, invK, self.height, self.width)   # differentiable in the depth map


class Project3D(nn.Module):
    """Holds `eps` (reference geometry_utils.py:66-70)."""

    def __init__(self, eps: float = 1e-8):
        super().__init__()
        self.register_buffer("eps", torch.tensor(eps).view(1, 1, 1))
        self._eps = float(eps)

    def forward(self, points_b4N, K_b44, cam_T_world_b44):
        """points [B,4,N] -> [B,3,N] = (pixel x, pixel y, depth + eps) (reference geometry_utils.py:72-89)."""
        pts, K, T = _diff("points_b4N", points_b4N), _data("K_b44", K_b44), _data("cam_T_world_b44", cam_T_world_b44)
        b, four, n = pts.shape
        if four != 4 or tuple(K.shape) != (b, 4, 4) or tuple(T.shape) != (b, 4, 4):
            raise ValueError("expected points [B,4,N], K [B,4,4], cam_T_world [B,4,4]")
        return _Project.apply(pts, K, T, self._eps)   # differentiable in the points


def pose_distance(pose_b44):
    """DVMVS pose distance (reference geometry_utils.py:178-191): (combined, R_measure, t_measure), each [B] -- the very
    values the metadata-MLP sweep computes for its pose channels (csrc/sr_dot_volume.hip: sr_geom_kernel).

    HOST tensors take the host path: the reference calls this inside dataset workers on CPU poses to order the source
    frames (generic_mvs_dataset.py:643-659) -- host-side data preparation like keyframes.py, a few scalar operations per
    pose in numpy (float32, the reference's operation order), not a fallback of the device hot path."""
    if isinstance(pose_b44, torch.Tensor) and not pose_b44.is_cuda:
        import numpy as np
        T = pose_b44.detach().to(torch.float32).numpy()
        R = T[:, :3, :3]
        tr = np.minimum(np.float32(3.0), R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2])
        r_m = np.sqrt(np.float32(2.0) * (np.float32(1.0) - tr / np.float32(3.0)))
        t_m = np.sqrt((T[:, :3, 3] ** 2).sum(1, dtype=np.float32))
        d = np.sqrt(t_m * t_m + r_m * r_m)
        return tuple(torch.from_numpy(np.ascontiguousarray(v.astype(np.float32))) for v in (d, r_m, t_m))
    T = _f32c("pose_b44", pose_b44)
    n = T.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=T.device)
    with _lib.on_device(T.device):
        rc = _lib.lib().sr_pose_distance_fwd(_lib.ptr(T), _lib.ptr(out), n, _lib.stream_ptr(T.device))
    _lib.check(rc, "sr_pose_distance_fwd")
    return out[:, 0], out[:, 1], out[:, 2]


def get_camera_rays(world_T_cam_b44, world_points_b3N, in_camera_frame, cam_T_world_b44=None, eps=1e-4):
    """Unit rays from the camera centre to the points (reference geometry_utils.py:143-175; `eps` is unused there too)."""
    pts = _f32c("world_points_b3N", world_points_b3N)
    T = _f32c("cam_T_world_b44" if in_camera_frame else "world_T_cam_b44",
              cam_T_world_b44 if in_camera_frame else world_T_cam_b44)
    b, three, n = pts.shape
    if three != 3 or tuple(T.shape) != (b, 4, 4):
        raise ValueError("expected points [B,3,N] and a [B,4,4] pose")
    out = torch.empty_like(pts)
    with _lib.on_device(pts.device):
        rc = _lib.lib().sr_camera_rays_fwd(_lib.ptr(pts), _lib.ptr(T), _lib.ptr(out), b, n, int(bool(in_camera_frame)),
                                           _lib.stream_ptr(pts.device))
    _lib.check(rc, "sr_camera_rays_fwd")
    return out
