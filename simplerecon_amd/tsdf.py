"""TSDF fusion of predicted depth maps -- API-compatible with the reference's tools/tsdf.py (`TSDF`, `TSDFFuser`)
and tools/fusers_helper.py (`OurFuser`), the step right after the depth hot path in `test.py --run_fusion`
(SURVEY.md §8f "next" #2).

`TSDFFuser.integrate_depth` runs ONE hand-written HIP kernel per batch of depth maps (`sr_tsdf_integrate_fwd`,
csrc/sr_tsdf.hip): the fp16 volume is streamed once, every voxel applies the frames of the batch in order in
registers.  The reference instead materialises ~10 [B, N]-sized fp16 temporaries per call (N = voxels) and does a
masked gather / scatter per frame.  Results are bit-identical to the reference executed on CPU (tests/golden/tsdf_*).

Not provided: mesh extraction (`to_mesh` / `save` need scikit-image marching cubes + trimesh; out of scope) and the
open3d fuser.  There is no CPU fallback: tensors must live on the GPU.
"""
import ctypes as C
from typing import Tuple

import numpy as np
import torch

from . import _lib


class TSDF:
    """Housing for a TSDF volume (reference tools/tsdf.py:11-97).  `tsdf_values`, `tsdf_weights`: fp16 [X,Y,Z];
    `voxel_coords`: fp16 [3,X,Y,Z] world coordinates (materialised lazily -- the HIP kernel regenerates them from
    `origin` and `voxel_size` when the volume comes from `from_bounds`)."""
    VOX_MOD = 8  # volume dimensions are multiples of 8 (tsdf.py:16)

    def __init__(self, voxel_coords, tsdf_values, tsdf_weights, voxel_size, origin):
        self._voxel_coords = voxel_coords.half() if voxel_coords is not None else None
        self.tsdf_values = tsdf_values.half().contiguous()
        self.tsdf_weights = tsdf_weights.half().contiguous()
        self.voxel_size = voxel_size
        self.origin = origin.half()
        self._origin_f32 = None   # set by from_bounds: the fp32 origin the coordinates are generated from
        self._generated = False   # voxel coordinates follow origin + index * voxel_size exactly

    @classmethod
    def from_bounds(cls, bounds: dict, voxel_size: float, device="cuda"):
        """Creates a TSDF volume with bounds at a specific voxel size (tsdf.py:69-97): values -1, weights 0."""
        expected_keys = ['xmin', 'xmax', 'ymin', 'ymax', 'zmin', 'zmax']
        for key in expected_keys:
            if key not in bounds.keys():
                raise KeyError("Provided bounds dict need to have keys"
                               "'xmin', 'xmax', 'ymin', 'ymax', 'zmin', 'zmax'!")
        dims = tuple(int(np.ceil((bounds[a + 'max'] - bounds[a + 'min']) / voxel_size / cls.VOX_MOD)) * cls.VOX_MOD
                     for a in "xyz")
        origin = torch.FloatTensor([bounds['xmin'], bounds['ymin'], bounds['zmin']])
        values = -torch.ones(dims, dtype=torch.float16, device=device)
        weights = torch.zeros(dims, dtype=torch.float16, device=device)
        vol = cls(None, values, weights, voxel_size, origin)
        vol._origin_f32 = origin.clone()
        vol._generated = True
        return vol

    @classmethod
    def generate_voxel_coords(cls, origin: torch.Tensor, volume_dims: Tuple[int, int, int], voxel_size: float):
        """World coordinates of every voxel, fp32 [3,X,Y,Z] (tsdf.py:99-111)."""
        grid = torch.meshgrid([torch.arange(vd, device=origin.device) for vd in volume_dims], indexing="ij")
        return origin.view(3, 1, 1, 1) + torch.stack(grid, 0) * voxel_size

    @property
    def voxel_coords(self):
        if self._voxel_coords is None:
            dev = self.tsdf_values.device
            self._voxel_coords = self.generate_voxel_coords(self._origin_f32.to(dev), tuple(self.tsdf_values.shape),
                                                            self.voxel_size).half()
        return self._voxel_coords

    def cuda(self):
        self.tsdf_values = self.tsdf_values.cuda()
        self.tsdf_weights = self.tsdf_weights.cuda()
        if self._voxel_coords is not None:
            self._voxel_coords = self._voxel_coords.cuda()

    def cpu(self):
        """Moves the volume to host memory (for export); `integrate_depth` needs it on the GPU again."""
        self.tsdf_values = self.tsdf_values.cpu()
        self.tsdf_weights = self.tsdf_weights.cpu()
        if self._voxel_coords is not None:
            self._voxel_coords = self._voxel_coords.cpu()

    def to_mesh(self, scale_to_world=True, export_single_mesh=False):
        raise NotImplementedError("mesh extraction (skimage marching cubes + trimesh, reference tsdf.py:128-160) is "
                                  "outside this path: take `tsdf_values` / `origin` / `voxel_size` to the reference's "
                                  "TSDF.to_mesh")

    def save(self, savepath, filename, save_mesh=True):
        raise NotImplementedError("see to_mesh")


class TSDFFuser:
    """Fuses depth maps into a TSDF volume (reference tools/tsdf.py:176-320)."""

    def __init__(self, tsdf, min_depth=0.5, max_depth=5.0, use_gpu=True):
        if not use_gpu:
            raise _lib.HipLibraryError("the HIP fuser has no CPU path (use_gpu=False)")
        self.tsdf = tsdf
        self.min_depth = min_depth
        self.max_depth = max_depth
        self.use_gpu = use_gpu
        self.truncation_size = 3.0
        self.maxW = 100.0

    @property
    def voxel_coords(self):
        return self.tsdf.voxel_coords

    @property
    def tsdf_values(self):
        return self.tsdf.tsdf_values

    @property
    def tsdf_weights(self):
        return self.tsdf.tsdf_weights

    @property
    def voxel_size(self):
        return self.tsdf.voxel_size

    @property
    def shape(self):
        return self.tsdf.tsdf_values.shape

    @property
    def truncation(self):
        return self.truncation_size * self.voxel_size

    def integrate_depth(self, depth_b1hw, cam_T_world_T_b44, K_b44, depth_mask_b1hw=None):
        """Integrates depth maps into the volume, frame after frame (tsdf.py:238-320).

        depth_b1hw: fp16 depth maps; cam_T_world_T_b44: fp16 extrinsics (not poses!); K_b44: fp16 intrinsics;
        depth_mask_b1hw: optional boolean mask of valid depth pixels.  The reference's arithmetic is fp16 throughout
        (its voxel coordinates are fp16, so fp32 inputs fail in its matmul); fp32 inputs are refused here as well."""
        vol = self.tsdf
        for name, t in (("depth_b1hw", depth_b1hw), ("cam_T_world_T_b44", cam_T_world_T_b44), ("K_b44", K_b44)):
            if not isinstance(t, torch.Tensor):
                raise TypeError(f"{name} must be a torch.Tensor")
            if t.dtype != torch.float16:
                raise TypeError(f"{name} must be float16 (the reference fuses in fp16: pass .half()), got {t.dtype}")
        dev = vol.tsdf_values.device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("the TSDF volume lives on the host: call tsdf.cuda() (no CPU fallback)")
        if depth_b1hw.dim() != 4 or depth_b1hw.shape[1] != 1:
            raise ValueError(f"depth_b1hw must be [B,1,H,W], got {tuple(depth_b1hw.shape)}")
        b, _, h, w = depth_b1hw.shape
        if tuple(cam_T_world_T_b44.shape) != (b, 4, 4) or tuple(K_b44.shape) != (b, 4, 4):
            raise ValueError("cam_T_world_T_b44 and K_b44 must be [B,4,4]")
        depth = depth_b1hw.to(dev).contiguous()   # the reference moves its inputs to the GPU as well (:257-260)
        T = cam_T_world_T_b44.to(dev).contiguous()
        K = K_b44.to(dev).contiguous()
        mask = None
        if depth_mask_b1hw is not None:
            if depth_mask_b1hw.dtype != torch.bool or tuple(depth_mask_b1hw.shape) != tuple(depth_b1hw.shape):
                raise ValueError("depth_mask_b1hw must be a boolean tensor shaped like depth_b1hw")
            mask = depth_mask_b1hw.to(dev).contiguous()
        X, Y, Z = vol.tsdf_values.shape
        if not (vol.tsdf_values.is_contiguous() and vol.tsdf_weights.is_contiguous()):
            raise ValueError("tsdf_values / tsdf_weights must be contiguous")
        if vol._generated:
            coords, o = None, vol._origin_f32
            ox, oy, oz = float(o[0]), float(o[1]), float(o[2])
        else:
            coords = vol.voxel_coords.to(dev).contiguous()
            ox = oy = oz = 0.0
        if b == 0:
            return
        lib = _lib.lib()
        f = C.c_float
        with torch.cuda.device(dev):
            rc = lib.sr_tsdf_integrate_fwd(
                _lib.ptr(vol.tsdf_values), _lib.ptr(vol.tsdf_weights), _lib.ptr(coords), X, Y, Z, f(ox), f(oy), f(oz),
                f(vol.voxel_size), _lib.ptr(depth), _lib.ptr(mask), _lib.ptr(K), _lib.ptr(T), b, h, w,
                f(self.min_depth), f(self.max_depth), f(self.max_depth - self.min_depth), f(self.truncation),
                f(self.maxW), _lib.stream_ptr(dev))
        _lib.check(rc, "sr_tsdf_integrate_fwd")


class OurFuser:
    """The fuser behind `--depth_fuser ours` (reference tools/fusers_helper.py:25-83) without mesh I/O: a dense TSDF
    over the given bounds (default: the reference's +-10 m cube when no ground-truth mesh limits the extent)."""

    def __init__(self, gt_path=None, fusion_resolution=0.04, max_fusion_depth=3, fuse_color=False, bounds=None,
                 device="cuda"):
        if gt_path:
            raise NotImplementedError("bounds from a ground-truth mesh need trimesh; pass bounds=dict(xmin=..., ...)")
        self.fusion_resolution = fusion_resolution
        self.max_fusion_depth = max_fusion_depth
        if bounds is None:
            bounds = dict(xmin=-10.0, xmax=10.0, ymin=-10.0, ymax=10.0, zmin=-10.0, zmax=10.0)
        tsdf_pred = TSDF.from_bounds(bounds, voxel_size=fusion_resolution, device=device)
        self.tsdf_fuser_pred = TSDFFuser(tsdf_pred, max_depth=max_fusion_depth)

    def fuse_frames(self, depths_b1hw, K_b44, cam_T_world_b44, color_b3hw=None):
        self.tsdf_fuser_pred.integrate_depth(depth_b1hw=depths_b1hw.half(), cam_T_world_T_b44=cam_T_world_b44.half(),
                                             K_b44=K_b44.half())
