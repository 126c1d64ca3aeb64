"""Geometry helpers of the hot path, API-compatible with reference utils/geometry_utils.py.

Only state (buffers with the reference's names, so checkpoints load with strict=True) and
the tiny host-side helpers live here; the per-pixel arithmetic of BackprojectDepth /
Project3D (reference geometry_utils.py:51-59, 72-89) is fused into the HIP sweep kernels."""
import torch
from torch import nn


class BackprojectDepth(nn.Module):
    """Holds `pix_coords_13N` (pixel centres at +0.5, reference geometry_utils.py:34-48)."""

    def __init__(self, height: int, width: int):
        super().__init__()
        self.height, self.width = height, width
        ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
        pix = torch.stack([xs.flatten() + 0.5, ys.flatten() + 0.5, torch.ones(height * width)], 0)
        self.register_buffer("pix_coords_13N", pix.unsqueeze(0).float())

    def forward(self, depth_b1hw, invK_b44):
        """Reference-semantics back-projection (geometry_utils.py:51-59) -- used by callers outside
        the fused kernels (it is plain torch; works on any device)."""
        cam = torch.matmul(invK_b44[:, :3, :3], self.pix_coords_13N)
        cam = depth_b1hw.flatten(start_dim=2) * cam
        return torch.cat([cam, torch.ones_like(cam[:, :1])], 1)


class Project3D(nn.Module):
    """Holds `eps` (reference geometry_utils.py:66-70)."""

    def __init__(self, eps: float = 1e-8):
        super().__init__()
        self.register_buffer("eps", torch.tensor(eps).view(1, 1, 1))

    def forward(self, points_b4N, K_b44, cam_T_world_b44):
        """Reference-semantics projection (geometry_utils.py:72-89), plain torch."""
        P = K_b44 @ cam_T_world_b44
        cam = P[:, :3] @ points_b4N
        z = cam[:, 2:]
        depth = z + self.eps
        scale = torch.where(z.abs() > self.eps, 1.0 / depth, torch.ones_like(depth))
        return torch.cat([cam[:, :2] * scale, depth], 1)


def pose_distance(pose_b44):
    """DVMVS pose distance (reference geometry_utils.py:178-191): (combined, R_measure, t_measure)."""
    R = pose_b44[:, :3, :3]
    t = pose_b44[:, :3, 3]
    tr = R.diagonal(offset=0, dim1=-1, dim2=-2).sum(-1)
    r_m = torch.sqrt(2 * (1 - torch.minimum(torch.ones_like(tr) * 3.0, tr) / 3))
    t_m = torch.norm(t, dim=1)
    return torch.sqrt(t_m ** 2 + r_m ** 2), r_m, t_m


def get_camera_rays(world_T_cam_b44, world_points_b3N, in_camera_frame, cam_T_world_b44=None, eps=1e-4):
    """Unit rays from the camera centre to the points (reference geometry_utils.py:143-175); plain torch
    helper kept for API parity -- inside the feature volume the rays are computed in the HIP sweep."""
    if in_camera_frame:
        ones = torch.ones_like(world_points_b3N[:, :1])
        rays = torch.matmul(cam_T_world_b44[:, :3, :4], torch.cat([world_points_b3N, ones], 1))
    else:
        rays = world_points_b3N - world_T_cam_b44[:, 0:3, 3][:, :, None]
    return torch.nn.functional.normalize(rays, dim=1)
