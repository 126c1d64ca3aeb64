"""Plane-sweep cost volumes -- drop-in for reference modules/cost_volume.py, backed by HIP.

Same class names, constructor arguments, `forward` keyword arguments, return tuples and
parameter/buffer names as the reference (cost_volume.py:13-380 `CostVolumeManager`,
:383-746 `FeatureVolumeManager`, :749-1164 `FastFeatureVolumeManager`), so that

    model.cost_volume = simplerecon_amd.cost_volume.to_hip(model.cost_volume)

works exactly like the reference's own `to_fast()` seam (reference test.py:196-198) and a
reference checkpoint loads with strict=True.  The arithmetic runs in hand-written gfx950
kernels reached through the C ABI of include/simplerecon_hip.h; there is no torch/CPU
fallback (inputs must be fp32 device tensors).  Both cost volumes have HIP backward kernels behind
torch.autograd.Functions (SURVEY.md §8f "next" #3), taken whenever grad mode is on and an input or parameter requires
grad -- like any torch module: the dot-product `CostVolumeManager` w.r.t. the matching features
(`_DotVolumeFunction`), the metadata-MLP `FeatureVolumeManager` w.r.t. the matching features and the six MLP tensors
(`_MlpVolumeFunction`).  `manager.differentiable = False` opts out (inputs that require grad are then refused).
"""
import ctypes as C

import torch
from torch import Tensor, nn

from . import _lib
from .geometry import BackprojectDepth, Project3D
from .networks import MLP


def _autocast_on(device_type):
    try:
        return torch.is_autocast_enabled(device_type)
    except TypeError:   # torch < 2.4: no device_type argument (the flag of the CUDA / HIP device)
        return torch.is_autocast_enabled()


def _autocast_to_f32(*tensors):
    """Inside a torch.autocast region the matching features reach the cost volume in fp16 / bf16 -- the reference trains
    with 16-bit autocast (options.py:100-101, train.py:132), where its grid_sample runs in fp32 and its MLP in half.
    The HIP kernels are fp32: upcast there (a differentiable torch cast, so gradients return in the caller's dtype).
    Outside autocast, non-fp32 inputs still fail loudly."""
    out = []
    for t in tensors:
        if isinstance(t, Tensor) and t.dtype in (torch.float16, torch.bfloat16) and _autocast_on(t.device.type):
            t = t.float()
        out.append(t)
    return out


class _DotVolumeFunction(torch.autograd.Function):
    """cost_volume, lowest_cost = sweep(cur_feats, src_feats; geometry) with the HIP backward
    (csrc/sr_dot_volume_bwd.hip) for the two feature tensors.  Geometry (poses, intrinsics, depth planes) is data, as
    in the reference, whose `lowest_cost` is computed under no_grad (cost_volume.py:360-372)."""

    @staticmethod
    def forward(ctx, mgr, cur, src, Ks, T, invK, planes):
        vol, lowest = mgr._launch_sweep(cur, src, Ks, T, invK, planes)
        ctx.save_for_backward(cur, src, Ks, T, invK, planes)
        ctx.mark_non_differentiable(lowest)
        return vol, lowest

    @staticmethod
    def backward(ctx, g_vol, _g_lowest):
        cur, src, Ks, T, invK, planes = ctx.saved_tensors
        need_cur, need_src = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if any(ctx.needs_input_grad[3:]):
            raise NotImplementedError("the cost volume is not differentiated w.r.t. poses / intrinsics / depth planes "
                                      "(they are data in the reference's training too)")
        b, k, c, h, w = src.shape
        d = planes.shape[1]
        dev = src.device
        lib = _lib.lib()
        g = g_vol if g_vol.dtype == torch.float32 else g_vol.float()
        if g.stride(2) != w * g.stride(3):
            g = g.contiguous()
        d_cur = torch.empty_like(cur) if need_cur else None
        d_src = torch.empty_like(src) if need_src else None
        if b == 0 or not (need_cur or need_src):
            return None, d_cur, d_src, None, None, None, None
        ws = torch.empty(lib.sr_volume_workspace_bytes(b, k, c, h, w), dtype=torch.uint8, device=dev)
        nscratch = lib.sr_dot_volume_bwd_scratch_bytes(b, k, c, h, w) if need_src else 0
        scratch = torch.empty(nscratch // 4, dtype=torch.float32, device=dev) if need_src else None
        st = _lib.stream_ptr(dev)
        with _lib.on_device(dev):
            # the forward's workspace may have been reused since: rebuild the geometry records + channels-last sources
            _lib.check(lib.sr_volume_prepare(_lib.ptr(src), _lib.ptr(Ks), _lib.ptr(T), None, b, k, c, h, w, _lib.ptr(ws),
                                             ws.numel(), st), "sr_volume_prepare")
            rc = lib.sr_dot_volume_bwd(_lib.ptr(g), g.stride(0), g.stride(1), g.stride(3), _lib.ptr(cur), _lib.ptr(invK),
                                       _lib.ptr(planes), *planes.stride(), b, k, c, h, w, d, _lib.ptr(d_cur),
                                       _lib.ptr(d_src), _lib.ptr(ws), ws.numel(), _lib.ptr(scratch), nscratch, st)
        _lib.check(rc, "sr_dot_volume_bwd")
        return None, d_cur, d_src, None, None, None, None


class _MlpVolumeFunction(torch.autograd.Function):
    """cost_volume, lowest_cost, mask = metadata-MLP sweep with the HIP backward (csrc/sr_mlp_volume_bwd.hip) for the two
    feature tensors and the six MLP tensors; geometry is data, `lowest_cost` / mask carry no gradient (as in the
    reference, cost_volume.py:360-372)."""

    @staticmethod
    def forward(ctx, mgr, return_mask, cur, src, Ks, T, Tp, invK, planes, W1, b1, W2, b2, W3, b3):
        vol, lowest, mask = mgr._launch_sweep(cur, src, Ks, T, Tp, invK, planes, (W1, b1, W2, b2, W3, b3), return_mask)
        ctx.save_for_backward(cur, src, Ks, T, Tp, invK, planes, W1, b1, W2, b2, W3, b3)
        if mask is not None:
            ctx.mark_non_differentiable(lowest, mask)   # one call: a second one would replace the first
            return vol, lowest, mask
        ctx.mark_non_differentiable(lowest)
        return vol, lowest, None

    @staticmethod
    def backward(ctx, g_vol, _g_lowest, _g_mask):
        cur, src, Ks, T, Tp, invK, planes, W1, b1, W2, b2, W3, b3 = ctx.saved_tensors
        if any(ctx.needs_input_grad[4:9]):
            raise NotImplementedError("the cost volume is not differentiated w.r.t. poses / intrinsics / depth planes "
                                      "(they are data in the reference's training too)")
        b, k, c, h, w = src.shape
        d = planes.shape[1]
        hidden = W1.shape[0]
        dev = src.device
        lib = _lib.lib()
        g = g_vol if g_vol.dtype == torch.float32 else g_vol.float()
        if g.stride(2) != w * g.stride(3):
            g = g.contiguous()
        d_cur, d_src = torch.empty_like(cur), torch.empty_like(src)
        grads = [torch.empty_like(t) for t in (W1, b1, W2, b2, W3, b3)]
        if b == 0:
            for t in grads:
                t.zero_()
        else:
            ws = torch.empty(lib.sr_volume_workspace_bytes(b, k, c, h, w), dtype=torch.uint8, device=dev)
            scratch = torch.empty(lib.sr_mlp_volume_bwd_scratch_bytes(b, k, c, h, w, hidden), dtype=torch.uint8,
                                  device=dev)
            st = _lib.stream_ptr(dev)
            with _lib.on_device(dev):
                _lib.check(lib.sr_volume_prepare(_lib.ptr(src), _lib.ptr(Ks), _lib.ptr(T), _lib.ptr(Tp), b, k, c, h, w,
                                                 _lib.ptr(ws), ws.numel(), st), "sr_volume_prepare")
                rc = lib.sr_mlp_volume_bwd(_lib.ptr(g), g.stride(0), g.stride(1), g.stride(3), _lib.ptr(cur),
                                           _lib.ptr(invK), _lib.ptr(planes), *planes.stride(),
                                           *[_lib.ptr(t.detach()) for t in (W1, b1, W2, b2, W3)], C.c_float(0.01), b, k, c,
                                           h, w, d, hidden, _lib.ptr(d_cur), _lib.ptr(d_src),
                                           *[_lib.ptr(t) for t in grads], _lib.ptr(ws), ws.numel(), _lib.ptr(scratch),
                                           scratch.numel(), st)
            _lib.check(rc, "sr_mlp_volume_bwd")
        need = ctx.needs_input_grad
        out = [None, None, d_cur if need[2] else None, d_src if need[3] else None, None, None, None, None, None]
        out += [gr if need[9 + i] else None for i, gr in enumerate(grads)]
        return tuple(out)


class CostVolumeManager(nn.Module):
    """Dot-product plane-sweep volume (reference cost_volume.py:13-380)."""

    differentiable = True   # HIP backward for cur_feats / src_feats (and, in the MLP managers, the six MLP tensors)

    def __init__(self, matching_height, matching_width, num_depth_bins=64, matching_dim_size=None,
                 num_source_views=None):
        super().__init__()
        self.num_depth_bins = num_depth_bins
        self.matching_height = matching_height
        self.matching_width = matching_width
        # memory format of the returned volume: contiguous (= the reference's b,d,h,w) or
        # channels_last (what the HIP CVEncoder consumes without a transpose)
        self.volume_memory_format = torch.contiguous_format
        self._workspace = None
        self.initialise_for_projection()

    # -- state with the reference's names (cost_volume.py:58-74) ---------------------------
    def initialise_for_projection(self):
        ramp = torch.linspace(0, 1, self.num_depth_bins).view(1, self.num_depth_bins, 1, 1)
        self.register_buffer("linear_ramp_1d11", ramp)
        self.backprojector = BackprojectDepth(height=self.matching_height, width=self.matching_width)
        self.projector = Project3D()

    # -- public helpers of the reference ----------------------------------------------------
    def get_mask(self, pix_coords_bk2hw):
        """Bounds mask of sampling locations (reference cost_volume.py:77-97)."""
        x, y = pix_coords_bk2hw[:, :, 0], pix_coords_bk2hw[:, :, 1]
        return (x > 2) & (x < self.matching_width - 2) & (y > 2) & (y < self.matching_height - 2)

    def generate_depth_planes(self, batch_size: int, min_depth: Tensor, max_depth: Tensor) -> Tensor:
        """Log-spaced planes as an expanded (stride-0) b,d,h,w view (reference cost_volume.py:100-136)."""
        ramp = self.linear_ramp_1d11.expand(batch_size, self.num_depth_bins, 1, 1)
        planes_bd11 = torch.exp(torch.log(min_depth) + torch.log(max_depth / min_depth) * ramp)
        if planes_bd11.stride(0) == 0:  # keep per-batch values addressable with plain strides
            planes_bd11 = planes_bd11.contiguous()
        return planes_bd11.expand(batch_size, self.num_depth_bins, self.matching_height, self.matching_width)

    def indices_to_disparity(self, indices, depth_planes_bdhw):
        """planes[argmax] lookup (reference cost_volume.py:338-342)."""
        return torch.gather(depth_planes_bdhw, dim=1, index=indices.unsqueeze(1)).squeeze(1)

    # -- HIP plumbing -------------------------------------------------------------------------
    def _check_inputs(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK):
        for name, t in (("cur_feats", cur_feats), ("src_feats", src_feats), ("src_extrinsics", src_extrinsics),
                        ("src_Ks", src_Ks), ("cur_invK", cur_invK)):
            _lib.require_device_f32(name, t)
        if self.differentiable:
            _lib.refuse_autograd(src_extrinsics, src_poses, src_Ks, cur_invK)
        else:
            _lib.refuse_autograd(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK)
        if src_feats.dim() != 5 or cur_feats.dim() != 4:
            raise ValueError("expected cur_feats [b,c,h,w] and src_feats [b,k,c,h,w]")
        b, k, c, h, w = src_feats.shape
        if (h, w) != (self.matching_height, self.matching_width):
            raise ValueError(f"feature maps are {h}x{w}, manager was built for "
                             f"{self.matching_height}x{self.matching_width}")
        if tuple(cur_feats.shape) != (b, c, h, w):
            raise ValueError(f"cur_feats {tuple(cur_feats.shape)} does not match src_feats {tuple(src_feats.shape)}")
        for name, t, shp in (("src_extrinsics", src_extrinsics, (b, k, 4, 4)), ("src_Ks", src_Ks, (b, k, 4, 4)),
                             ("cur_invK", cur_invK, (b, 4, 4))):
            if tuple(t.shape) != shp:
                raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shp}")
        return b, k, c, h, w

    def _planes(self, batch_size, min_depth, max_depth, depth_planes_bdhw):
        if depth_planes_bdhw is None:
            depth_planes_bdhw = self.generate_depth_planes(batch_size, min_depth, max_depth)
        _lib.require_device_f32("depth_planes_bdhw", depth_planes_bdhw)
        exp = (batch_size, self.num_depth_bins, self.matching_height, self.matching_width)
        if tuple(depth_planes_bdhw.shape) != exp:
            raise ValueError(f"depth_planes_bdhw has shape {tuple(depth_planes_bdhw.shape)}, expected {exp}")
        return depth_planes_bdhw

    def _get_workspace(self, nbytes, device):
        """Sweep workspace (geometry records, channels-last source features, packed MLP weights, argmax keys).  One
        buffer per (device, HIP stream): sub-batches on different streams (DepthModel.hot_path, num_streams > 1) must
        not share it.  Inside a HIP-graph capture a FRESH buffer is allocated from the graph's private pool instead:
        the graph bakes the pointer in and owns the memory, so later eager calls that grow (and drop) the cached
        buffer cannot leave a replay writing into freed memory."""
        if _lib.capturing():
            return torch.empty(nbytes, dtype=torch.uint8, device=device)
        if self._workspace is None:
            self._workspace = {}
        key = (device, _lib.stream_id(device))
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._workspace[key] = ws
        return ws

    def _alloc_outputs(self, b, h, w, device, return_mask):
        d = self.num_depth_bins
        vol = torch.empty((b, d, h, w), dtype=torch.float32, device=device,
                          memory_format=self.volume_memory_format)
        lowest = torch.empty((b, h, w), dtype=torch.float32, device=device)
        mask = torch.empty((b, h, w), dtype=torch.uint8, device=device) if return_mask else None
        return vol, lowest, mask

    @staticmethod
    def _volume_strides(vol):
        sb, sd, sy, sx = vol.stride()
        if sy != vol.shape[3] * sx:
            raise ValueError("cost volume rows must be dense")
        return sb, sd, sx

    def _sweep(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
               depth_planes_bdhw, return_mask):
        cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK = _autocast_to_f32(
            cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK)
        b, k, c, h, w = self._check_inputs(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK)
        planes = self._planes(b, min_depth, max_depth, depth_planes_bdhw)
        cur, src = cur_feats.contiguous(), src_feats.contiguous()
        Ks, T, invK = src_Ks.contiguous(), src_extrinsics.contiguous(), cur_invK.contiguous()
        if torch.is_grad_enabled() and (cur.requires_grad or src.requires_grad):
            _lib.refuse_autograd(planes)
            vol, lowest = _DotVolumeFunction.apply(self, cur, src, Ks, T, invK, planes)
        else:
            vol, lowest = self._launch_sweep(cur, src, Ks, T, invK, planes)
        # the dot model ignores return_mask and returns None (reference cost_volume.py:286, 335)
        return vol, lowest, planes, None

    def _launch_sweep(self, cur, src, Ks, T, invK, planes):
        b, k, c, h, w = src.shape
        dev = src.device
        lib = _lib.lib()
        vol, lowest, _ = self._alloc_outputs(b, h, w, dev, False)
        if b == 0:
            return vol, lowest
        nws = lib.sr_volume_workspace_bytes(b, k, c, h, w)
        ws = self._get_workspace(nws, dev)
        sb, sd, sp = self._volume_strides(vol)
        with _lib.on_device(dev):
            rc = lib.sr_dot_volume_fwd(
                _lib.ptr(cur), _lib.ptr(src), _lib.ptr(Ks), _lib.ptr(T), _lib.ptr(invK), _lib.ptr(planes),
                *planes.stride(), b, k, c, h, w, self.num_depth_bins, _lib.ptr(vol), sb, sd, sp,
                _lib.ptr(lowest), C.c_void_p(0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "sr_dot_volume_fwd")
        return vol, lowest

    def _warp(self, src_feats, src_extrinsics, src_Ks, cur_invK, planes, want_pix):
        """Runs sr_warp_features_fwd for the Dp planes of `planes` ([b,Dp,h,w], any strides)."""
        for name, t in (("src_feats", src_feats), ("src_extrinsics", src_extrinsics), ("src_Ks", src_Ks),
                        ("cur_invK", cur_invK), ("depth planes", planes)):
            _lib.require_device_f32(name, t)
        _lib.refuse_autograd(src_feats, src_extrinsics, src_Ks, cur_invK, planes)
        b, k, c, h, w = src_feats.shape
        dp = planes.shape[1]
        dev = src_feats.device
        n = h * w
        world = torch.empty((b, dp, 4, n), dtype=torch.float32, device=dev)
        depths = torch.empty((b, k, dp, h, w), dtype=torch.float32, device=dev)
        warped = torch.empty((b, k, dp, c, h, w), dtype=torch.float32, device=dev)
        mask = torch.empty((b, k, dp, h, w), dtype=torch.float32, device=dev)
        pix = torch.empty((b, k, dp, 2, h, w), dtype=torch.float32, device=dev) if want_pix else None
        if b == 0:
            return world, depths, warped, mask, pix
        lib = _lib.lib()
        ws = self._get_workspace(lib.sr_volume_workspace_bytes(b, k, c, h, w), dev)
        with _lib.on_device(dev):
            rc = lib.sr_warp_features_fwd(
                _lib.ptr(src_feats.contiguous()), _lib.ptr(src_Ks.contiguous()), _lib.ptr(src_extrinsics.contiguous()),
                _lib.ptr(cur_invK.contiguous()), _lib.ptr(planes), *planes.stride(), b, k, c, h, w, dp,
                _lib.ptr(world), _lib.ptr(depths), _lib.ptr(warped), _lib.ptr(mask), _lib.ptr(pix), _lib.ptr(ws),
                ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "sr_warp_features_fwd")
        return world, depths, warped, mask, pix

    def warp_features(self, src_feats, src_extrinsics, src_Ks, cur_invK, depth_plane_b1hw, batch_size,
                      num_src_frames, num_feat_channels, uv_scale=None):
        """One-plane warp with the reference's signature and return tuple (cost_volume.py:139-234):
        (world_points_B4N, depths_bkhw, src_feat_warped_bkchw, mask_bkhw).  `uv_scale` is implied by
        the feature-map size and accepted for signature compatibility only."""
        world, depths, warped, mask, _ = self._warp(src_feats, src_extrinsics, src_Ks, cur_invK, depth_plane_b1hw,
                                                    False)
        world_B4N = world[:, 0].repeat_interleave(num_src_frames, dim=0)
        return world_B4N, depths[:, :, 0], warped[:, :, 0], mask[:, :, 0]

    # -- the reference's entry points -------------------------------------------------------
    def build_cost_volume(self, cur_feats: Tensor, src_feats: Tensor, src_extrinsics: Tensor, src_poses: Tensor,
                          src_Ks: Tensor, cur_invK: Tensor, min_depth: Tensor, max_depth: Tensor,
                          depth_planes_bdhw: Tensor = None, return_mask: bool = False):
        """Returns (volume_bdhw, depth_planes_bdhw, overall_mask_bhw | None) like reference
        cost_volume.py:237-335 / 451-736."""
        vol, _, planes, mask = self._sweep(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK,
                                           min_depth, max_depth, depth_planes_bdhw, return_mask)
        return vol, planes, mask

    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                depth_planes_bdhw=None, return_mask=False):
        """Returns (cost_volume, lowest_cost, depth_planes_bdhw, overall_mask_bhw) like reference
        cost_volume.py:345-380; the argmax/gather is fused into the sweep kernel."""
        return self._sweep(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth,
                           max_depth, depth_planes_bdhw, return_mask)


def mlp_input_channels(matching_dim_size, num_source_views):
    """mlp_channels[0] of the reference (cost_volume.py:420-435): visual + depth + ray + angle +
    mask + dot + pose-penalty channels = C(1+K) + 10K + 4  (202 for C=16, K=7)."""
    k, c = num_source_views, matching_dim_size
    return c * (1 + k) + (1 + k) + 3 * (1 + k) + k + k + k + 3 * k


class FeatureVolumeManager(CostVolumeManager):
    """Metadata-MLP feature volume (reference cost_volume.py:383-746).

    `mlp_channels` is taken BY VALUE (the reference mutates a shared default list,
    cost_volume.py:402, 429 -- harmless there, not replicated)."""

    # The MLP sweep has a HIP backward (csrc/sr_mlp_volume_bwd.hip, gradients = the reference's autograd): a swapped-in
    # manager trains under the reference's train.py like the torch one it replaces.  As with any nn.Module, a call in
    # grad mode with parameters that require grad builds an autograd graph; inference runs under no_grad /
    # inference_mode (reference test.py:210).
    differentiable = True
    # CUs the NEXT sweep leaves to other streams (0: none).  Set and cleared by DepthModel around its own call, never persistent:
    # a persistent workgroup of the sweep owns its CU, so work on other streams is parked for the whole sweep otherwise.
    _reserve_cus = 0
    _reserve_oid = None

    def __init__(self, matching_height, matching_width, num_depth_bins=64, mlp_channels=(202, 128, 128, 1),
                 matching_dim_size=16, num_source_views=7):
        super().__init__(matching_height, matching_width, num_depth_bins)
        chans = list(mlp_channels)
        chans[0] = mlp_input_channels(matching_dim_size, num_source_views)
        self.matching_dim_size = matching_dim_size
        self.num_source_views = num_source_views
        self.mlp_channels = chans
        self.mlp = MLP(channel_list=chans, disable_final_activation=True)
        self._packed = None

    def _mlp_params(self):
        lin = [m for m in self.mlp.net if isinstance(m, nn.Linear)]
        if len(lin) != 3 or lin[2].out_features != 1 or lin[0].out_features != lin[1].in_features \
                or lin[1].out_features != lin[1].in_features:
            raise _lib.HipLibraryError(f"HIP matching MLP supports [Cin, H, H, 1] channel lists, got {self.mlp_channels}")
        return lin

    def _sweep(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
               depth_planes_bdhw, return_mask):
        cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK = _autocast_to_f32(
            cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK)
        b, k, c, h, w = self._check_inputs(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK)
        _lib.require_device_f32("src_poses", src_poses)
        if tuple(src_poses.shape) != (b, k, 4, 4):
            raise ValueError(f"src_poses has shape {tuple(src_poses.shape)}, expected {(b, k, 4, 4)}")
        lin = self._mlp_params()
        if lin[0].in_features != mlp_input_channels(c, k):
            raise ValueError(f"MLP expects {lin[0].in_features} input channels but {k} views x {c} channels "
                             f"give {mlp_input_channels(c, k)}")
        planes = self._planes(b, min_depth, max_depth, depth_planes_bdhw)
        cur, src = cur_feats.contiguous(), src_feats.contiguous()
        Ks, T, Tp, invK = (src_Ks.contiguous(), src_extrinsics.contiguous(), src_poses.contiguous(),
                           cur_invK.contiguous())
        weights = (lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias)
        for i, t in enumerate(weights):
            _lib.require_device_f32(f"mlp parameter {i}", t)
        wants_grad = self.differentiable and torch.is_grad_enabled() and \
            any(t.requires_grad for t in (cur, src) + weights)
        if wants_grad:
            _lib.refuse_autograd(planes)
            vol, lowest, mask = _MlpVolumeFunction.apply(self, bool(return_mask), cur, src, Ks, T, Tp, invK, planes,
                                                         *[t.contiguous() for t in weights])
        else:
            vol, lowest, mask = self._launch_sweep(cur, src, Ks, T, Tp, invK, planes,
                                                   [t.detach().contiguous() for t in weights], return_mask)
        return vol, lowest, planes, (mask.bool() if mask is not None else None)

    def _launch_sweep(self, cur, src, Ks, T, Tp, invK, planes, params, return_mask):
        b, k, c, h, w = src.shape
        dev = src.device
        lib = _lib.lib()
        vol, lowest, mask = self._alloc_outputs(b, h, w, dev, return_mask)
        if b == 0:
            return vol, lowest, mask
        params = [t.detach() for t in params]
        hidden = params[0].shape[0]
        nws = lib.sr_mlp_volume_workspace_bytes(b, k, c, h, w, hidden)
        ws = self._get_workspace(nws, dev)
        sb, sd, sp = self._volume_strides(vol)
        from . import ops   # (ops.PROFILE: bench.py's in-step kernel table)
        prof = ops.PROFILE
        reserve = int(getattr(self, "_reserve_cus", 0))
        with _lib.SPLIT_GUARD, _lib.on_device(dev):   # (SR_MLP_SPLIT is read by the packing and by the sweep inside this one call)
            if reserve > 0:
                # this call only: the persistent grid leaves `reserve` CUs to other streams (DepthModel: the image-prior encoder on
                # its side stream).  The option table is process-wide; every sweep launched through this module holds the guard.
                if FeatureVolumeManager._reserve_oid is None:
                    FeatureVolumeManager._reserve_oid = _lib._option_id("SR_MLP_RESERVE_CUS")
                prev_reserve = C.c_int(0)
                _lib.check(lib.sr_option_set(FeatureVolumeManager._reserve_oid, reserve, C.byref(prev_reserve)), "sr_option_set")
            if prof is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            rc = lib.sr_mlp_volume_fwd(
                _lib.ptr(cur), _lib.ptr(src), _lib.ptr(Ks), _lib.ptr(T), _lib.ptr(Tp), _lib.ptr(invK),
                _lib.ptr(planes), *planes.stride(), *[_lib.ptr(t) for t in params], hidden,
                C.c_float(0.01),  # nn.LeakyReLU default slope (reference networks.py:139)
                b, k, c, h, w, self.num_depth_bins, _lib.ptr(vol), sb, sd, sp, _lib.ptr(lowest),
                _lib.ptr(mask), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            if reserve > 0:
                lib.sr_option_set(FeatureVolumeManager._reserve_oid, prev_reserve.value, C.byref(C.c_int(0)))
            if prof is not None:
                ev1.record()
                cin = c * (k + 1) + 10 * k + 4
                prof.append(("sr_mlp_volume_fwd", 2.0 * (cin * hidden + hidden * hidden + hidden) * b * self.num_depth_bins * h * w,
                             ev0, ev1, (), None))
        _lib.check(rc, "sr_mlp_volume_fwd")
        return vol, lowest, mask

    def to_fast(self) -> "FastFeatureVolumeManager":
        """Same seam as the reference (cost_volume.py:739-746): shares the MLP."""
        manager = FastFeatureVolumeManager(self.matching_height, self.matching_width,
                                           num_depth_bins=self.num_depth_bins, mlp_channels=self.mlp_channels,
                                           matching_dim_size=self.matching_dim_size,
                                           num_source_views=self.num_source_views)
        manager.to(self.linear_ramp_1d11.device)
        manager.mlp = self.mlp
        manager.volume_memory_format = self.volume_memory_format
        return manager


class FastFeatureVolumeManager(FeatureVolumeManager):
    """The reference's batched variant (cost_volume.py:749-1164) trades O(B*D*N*202) memory for
    fewer launches.  The fused HIP sweep already is one launch with O(1) intermediates, so this
    class is the same kernel under the reference's second name."""

    def warp_features(self, src_feats, src_extrinsics, src_Ks, cur_invK, depth_plane_bdhw, batch_size,
                      num_src_frames, num_feat_channels, uv_scale=None):
        """All-planes warp with the reference's signature and return tuple (cost_volume.py:812-964):
        (world_points_bkd4hw, depths_bkdhw, src_feat_warped_bkdchw, mask_bkdhw, pix_coords_bkd2hw)."""
        world, depths, warped, mask, pix = self._warp(src_feats, src_extrinsics, src_Ks, cur_invK,
                                                      depth_plane_bdhw, True)
        b, dp = world.shape[:2]
        h, w = self.matching_height, self.matching_width
        world_bkd4hw = world.view(b, 1, dp, 4, h, w).expand(b, num_src_frames, dp, 4, h, w)
        return world_bkd4hw, depths, warped, mask, pix


def to_hip(manager):
    """Builds the HIP-backed twin of a REFERENCE manager instance (or of one of ours) -- the attribute-swap seam of
    reference test.py:196-198:

        model.cost_volume = simplerecon_amd.cost_volume.to_hip(model.cost_volume)

    Like the reference's own `to_fast()` (cost_volume.py:739-746) the twin SHARES the matching MLP module with the
    manager it was built from (same nn.Parameters: an optimizer / checkpoint that holds them keeps working); a
    `FastFeatureVolumeManager` comes back as a `FastFeatureVolumeManager`."""
    h, w, d = manager.matching_height, manager.matching_width, manager.num_depth_bins
    dev = manager.linear_ramp_1d11.device
    if hasattr(manager, "mlp"):
        lin = [m for m in manager.mlp.net if isinstance(m, nn.Linear)]
        cin = lin[0].in_features
        chans = [cin] + [m.out_features for m in lin]
        # C(1+K) + 10K + 4 = cin; recover (C, K) from the attributes when present, else assume C = 16
        c = getattr(manager, "matching_dim_size", 16)
        k = (cin - c - 4) // (c + 10)
        cls = FastFeatureVolumeManager if "Fast" in type(manager).__name__ else FeatureVolumeManager
        new = cls(h, w, num_depth_bins=d, mlp_channels=chans, matching_dim_size=c, num_source_views=k)
        new.to(dev)
        new.mlp = manager.mlp       # shared, not copied
    else:
        new = CostVolumeManager(h, w, num_depth_bins=d).to(dev)
    new.linear_ramp_1d11.copy_(manager.linear_ramp_1d11)
    if hasattr(manager, "volume_memory_format"):
        new.volume_memory_format = manager.volume_memory_format
    return new
