"""GPU tests of the remaining public API surface of the reference's hot-path classes:
warp_features (slow + fast signatures), MLP.forward, the K=15 / D=96 stress shape (BASELINE.json configs[4])."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_cases as gc
import oracle
from parity import assert_close, mismatch_fraction, rel_err
from simplerecon_amd import geometry, synthetic
from simplerecon_amd.cost_volume import CostVolumeManager, FastFeatureVolumeManager, FeatureVolumeManager
from simplerecon_amd.networks import MLP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_backproject(depth_b1hw, invK, h, w):
    """ATen statement of BackprojectDepth.forward (test-side reference)."""
    ys, xs = torch.meshgrid(torch.arange(h, device=DEV), torch.arange(w, device=DEV), indexing="ij")
    pix = torch.stack([xs.flatten() + 0.5, ys.flatten() + 0.5, torch.ones(h * w, device=DEV)], 0).unsqueeze(0).float()
    cam = depth_b1hw.flatten(start_dim=2) * torch.matmul(invK[:, :3, :3], pix)
    return torch.cat([cam, torch.ones_like(cam[:, :1])], 1)


def _torch_project(points_b4N, K, T, eps=1e-8):
    """ATen statement of Project3D.forward (test-side reference)."""
    cam = (K @ T)[:, :3] @ points_b4N
    z = cam[:, 2:]
    depth = z + eps
    scale = torch.where(z.abs() > eps, 1.0 / depth, torch.ones_like(depth))
    return torch.cat([cam[:, :2] * scale, depth], 1)


def test_geometry_helpers_on_hip():
    """BackprojectDepth / Project3D / pose_distance / get_camera_rays as standalone HIP calls (csrc/sr_geometry.hip)."""
    B, K, h, w = 2, 3, 10, 14
    inp = {k: v.to(DEV) for k, v in synthetic.cost_volume_inputs(B, K, 16, h, w, seed=4).items()}
    depth = 0.5 + 3.0 * torch.rand((B, 1, h, w), device=DEV)
    pts = geometry.BackprojectDepth(h, w).to(DEV)(depth, inp["cur_invK"])
    assert_close(pts, _torch_backproject(depth, inp["cur_invK"], h, w), tol=1e-6, what="BackprojectDepth")
    Ks, T = inp["src_Ks"][:, 0].contiguous(), inp["src_extrinsics"][:, 0].contiguous()
    cam = geometry.Project3D().to(DEV)(pts, Ks, T)
    assert_close(cam, _torch_project(pts, Ks, T), tol=1e-5, what="Project3D")
    poses = inp["src_poses"].view(-1, 4, 4)
    dist, rm, tm = geometry.pose_distance(poses)
    R, t = poses[:, :3, :3], poses[:, :3, 3]
    tr = R.diagonal(dim1=-2, dim2=-1).sum(-1)
    rm_r = torch.sqrt(2 * (1 - torch.clamp(tr, max=3.0) / 3))
    tm_r = t.norm(dim=1)
    # R_measure = sqrt(2 (1 - tr/3)) cancels catastrophically near the identity: compare squared values
    assert torch.allclose(tm, tm_r, rtol=1e-6) and torch.allclose(rm ** 2, rm_r ** 2, atol=2e-7)
    assert torch.allclose(dist ** 2, tm_r ** 2 + rm_r ** 2, rtol=1e-5, atol=2e-7)
    wpts = pts[:, :3].contiguous()
    rays = geometry.get_camera_rays(inp["src_poses"][:, 0].contiguous(), wpts, in_camera_frame=False)
    ref = F.normalize(wpts - inp["src_poses"][:, 0, :3, 3][:, :, None], dim=1)
    assert_close(rays, ref, tol=1e-6, what="get_camera_rays (world frame)")
    rays_c = geometry.get_camera_rays(None, wpts, in_camera_frame=True, cam_T_world_b44=T)
    ref_c = F.normalize(torch.matmul(T[:, :3, :4], torch.cat([wpts, torch.ones_like(wpts[:, :1])], 1)), dim=1)
    assert_close(rays_c, ref_c, tol=1e-6, what="get_camera_rays (camera frame)")
    # host poses (the reference's dataset workers, generic_mvs_dataset.py:643-659) take the host path: same values
    dist_h, rm_h, tm_h = geometry.pose_distance(poses.cpu())
    assert not dist_h.is_cuda and torch.allclose(tm_h, tm.cpu(), rtol=1e-6) and torch.allclose(rm_h ** 2, rm.cpu() ** 2, atol=2e-7)
    with pytest.raises(Exception):
        geometry.BackprojectDepth(h, w)(depth.cpu(), inp["cur_invK"].cpu())   # device modules: no CPU fallback


def test_geometry_modules_are_differentiable_like_the_reference():
    """The reference's multi-view losses differentiate BackprojectDepth in the depth map and Project3D in the points
    (geometry_utils.py:51-59, 72-89): the HIP adjoints equal torch autograd of the same composition; cameras are data."""
    B, K, h, w = 2, 3, 9, 13
    inp = {k: v.to(DEV) for k, v in synthetic.cost_volume_inputs(B, K, 16, h, w, seed=5).items()}
    Ks, T = inp["src_Ks"][:, 0].contiguous(), inp["src_extrinsics"][:, 0].contiguous()
    torch.manual_seed(3)
    depth0 = 0.5 + 3.0 * torch.rand((B, 1, h, w), device=DEV)
    cot = torch.randn((B, 3, h * w), device=DEV)
    bp, pr = geometry.BackprojectDepth(h, w).to(DEV), geometry.Project3D().to(DEV)

    depth = depth0.clone().requires_grad_(True)
    pts = bp(depth, inp["cur_invK"])
    cam = pr(pts, Ks, T)
    (cam * cot).sum().backward()

    depth_r = depth0.clone().requires_grad_(True)
    pts_r = _torch_backproject(depth_r, inp["cur_invK"], h, w)
    pts_r.retain_grad()
    cam_r = _torch_project(pts_r, Ks, T)
    (cam_r * cot).sum().backward()
    assert_close(cam.detach(), cam_r.detach(), tol=1e-5, what="forward under autograd")
    assert_close(depth.grad, depth_r.grad, tol=1e-5, what="d loss / d depth")

    pts_leaf = pts_r.detach().clone().requires_grad_(True)
    (pr(pts_leaf, Ks, T) * cot).sum().backward()
    assert_close(pts_leaf.grad, pts_r.grad, tol=1e-5, what="d loss / d points")
    with pytest.raises(Exception, match="cameras are data"):
        bp(depth0, inp["cur_invK"].clone().requires_grad_(True))


def _torch_warp(inp, planes_b1hw, h, w):
    """Plain PyTorch fp32 reference of the op (same composition as reference cost_volume.py:139-234)."""
    b, k, c = inp["src_feats"].shape[:3]
    wp = _torch_backproject(planes_b1hw, inp["cur_invK"], h, w).repeat_interleave(k, dim=0)
    cam = _torch_project(wp, inp["src_Ks"].view(-1, 4, 4), inp["src_extrinsics"].view(-1, 4, 4)).view(-1, 3, h, w)
    scale = torch.tensor([1 / w, 1 / h], device=DEV).view(1, 1, 1, 2)
    uv = 2 * cam[:, :2].permute(0, 2, 3, 1) * scale - 1
    warped = F.grid_sample(inp["src_feats"].view(-1, c, h, w), uv, padding_mode="zeros", mode="bilinear",
                           align_corners=False).view(b, k, c, h, w)
    depths = cam[:, 2:].view(b, k, h, w)
    return wp, depths, warped, (depths > 0).float(), cam[:, :2].view(b, k, 2, h, w)


@pytest.mark.parametrize("name", ["dot_small", "dot_edge"])
def test_warp_features_matches_torch_reference(name):
    case = gc.VOLUME_CASES[name]
    inp = {k: v.to(DEV) for k, v in gc.volume_inputs(case).items()}
    b, k, c, h, w, d = case["B"], case["K"], case["C"], case["h"], case["w"], case["D"]
    mgr = CostVolumeManager(h, w, num_depth_bins=d).to(DEV)
    planes = mgr.generate_depth_planes(b, inp["min_depth"], inp["max_depth"])
    with torch.inference_mode():
        for j in (0, d - 1):
            plane = planes[:, j].unsqueeze(1)
            wp, depths, warped, mask = mgr.warp_features(inp["src_feats"], inp["src_extrinsics"], inp["src_Ks"],
                                                         inp["cur_invK"], plane, b, k, c, None)
            wp_r, depths_r, warped_r, mask_r, _ = _torch_warp(inp, plane.contiguous(), h, w)
            assert wp.shape == wp_r.shape and warped.shape == (b, k, c, h, w)
            assert_close(wp, wp_r, tol=1e-6, what="world points")
            assert_close(depths, depths_r, tol=1e-5, what="depths")
            assert_close(warped, warped_r, tol=1e-4, what="warped features")
            assert mismatch_fraction(mask, mask_r) < 1e-3
        # consistency with the fused sweep: sum_k mask * sum_c warped * cur == volume plane
        vol = mgr(**inp)[0]
        plane = planes[:, d - 1].unsqueeze(1)
        _, _, warped, mask = mgr.warp_features(inp["src_feats"], inp["src_extrinsics"], inp["src_Ks"], inp["cur_invK"],
                                               plane, b, k, c, None)
        dot = ((warped * inp["cur_feats"].unsqueeze(1)).sum(2) * mask).sum(1)
        assert_close(dot, vol[:, d - 1], tol=2e-6, what="warp_features vs fused sweep")


def test_fast_warp_features_all_planes():
    case = gc.VOLUME_CASES["hero_small"]
    inp = {k: v.to(DEV) for k, v in gc.volume_inputs(case).items()}
    b, k, c, h, w, d = case["B"], case["K"], case["C"], case["h"], case["w"], case["D"]
    mgr = FastFeatureVolumeManager(h, w, num_depth_bins=d, num_source_views=k).to(DEV)
    planes = mgr.generate_depth_planes(b, inp["min_depth"], inp["max_depth"])
    with torch.inference_mode():
        wp, depths, warped, mask, pix = mgr.warp_features(inp["src_feats"], inp["src_extrinsics"], inp["src_Ks"],
                                                          inp["cur_invK"], planes, b, k, c, None)
        assert wp.shape == (b, k, d, 4, h, w) and depths.shape == (b, k, d, h, w)
        assert warped.shape == (b, k, d, c, h, w) and pix.shape == (b, k, d, 2, h, w)
        for j in (0, 3, d - 1):
            _, depths_r, warped_r, mask_r, pix_r = _torch_warp(inp, planes[:, j].unsqueeze(1).contiguous(), h, w)
            assert_close(depths[:, :, j], depths_r, tol=1e-5, what="depths")
            assert_close(warped[:, :, j], warped_r, tol=1e-4, what="warped")
            assert_close(pix[:, :, j], pix_r, tol=1e-4, what="pix coords")
        assert bool((mgr.get_mask(pix[:, :, -1]).any(1) & (mask[:, :, -1] > 0).any(1)).any())


def test_mlp_forward_hip():
    mlp = synthetic.seeded_fill_(MLP([202, 128, 128, 1], disable_final_activation=True), seed=5).to(DEV)
    x = torch.randn(3, 7, 11, 202, device=DEV)
    with torch.inference_mode():
        y = mlp(x)
        ref = mlp.net(x)  # plain PyTorch fp32 reference of the same op
    assert y.shape == (3, 7, 11, 1)
    assert_close(y, ref, tol=1e-5, what="MLP.forward")


def test_stress_shape_k15_d96():
    """BASELINE.json configs[4] channel layout: 15 source views (410-input MLP, W1 streamed from L2 because it
    exceeds the LDS), 96 planes -- at a reduced spatial size so the oracle finishes in seconds."""
    B, K, C, D, h, w = 1, 15, 16, 96, 18, 24
    inp = synthetic.cost_volume_inputs(B, K, C, h, w, seed=77)
    mgr = FeatureVolumeManager(h, w, num_depth_bins=D, matching_dim_size=C, num_source_views=K)
    assert mgr.mlp.net[0].in_features == 410
    synthetic.seeded_fill_(mgr.mlp, seed=9)
    mgr = mgr.to(DEV)
    with torch.inference_mode():
        vol, lowest, planes, mask = mgr(return_mask=True, **{k: v.to(DEV) for k, v in inp.items()})
    torch.cuda.synchronize()
    n = {k: v.numpy() for k, v in inp.items()}
    sd = {k: v.cpu().numpy() for k, v in mgr.mlp.state_dict().items()}
    mlp = dict(W1=sd["net.0.weight"], b1=sd["net.0.bias"], W2=sd["net.2.weight"], b2=sd["net.2.bias"],
               W3=sd["net.4.weight"], b3=sd["net.4.bias"])
    cv_o, low_o, mask_o = oracle.mlp_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"],
                                            n["src_poses"], n["cur_invK"], planes[:, :, 0, 0].cpu().numpy(), mlp,
                                            want_mask=True)
    assert_close(vol, cv_o, tol=2e-5, what="K=15, D=96 vs oracle")
    assert mismatch_fraction(mask, mask_o) == 0.0
    # dot model on the same stress shape
    dm = CostVolumeManager(h, w, num_depth_bins=D).to(DEV)
    with torch.inference_mode():
        dv = dm(**{k: v.to(DEV) for k, v in inp.items()})[0]
    dv_o = oracle.dot_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["cur_invK"],
                             planes[:, :, 0, 0].cpu().numpy())[0]
    assert_close(dv, dv_o, tol=2e-6, what="dot K=15, D=96 vs oracle")
