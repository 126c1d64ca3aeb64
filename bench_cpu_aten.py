"""ATen (PyTorch CPU, fp32) composition of the hot path -- the `cpu_baseline` leg of bench.py.

BASELINE.json's north star asks for "the reference's CPU PyTorch path timed on the same box's host cores".  The
reference itself cannot travel to the GPU box (and `DepthModel` needs pytorch_lightning / timm / antialiased_cnns,
which this image does not have), so this file states the SAME ATen operator sequence the reference executes on CPU --
`torch.matmul` back-projection / projection, `F.grid_sample(bilinear, zeros, align_corners=False)`, `F.normalize`,
`F.cosine_similarity`, `torch.cat`, `F.linear` + `F.leaky_relu` per depth plane (cost_volume.py:451-736), `F.conv2d`
BasicBlocks (layers.py:24-85), `F.interpolate` upsampling (generic_utils.py:96-105) -- on the weights of OUR modules
(same state-dict layout), with `torch.set_num_threads(all host cores)`.  It is a baseline only (kind "port"): nothing
here is used by the product path, and its numbers say nothing about kernel quality.

Checked against the oracle in tests/test_oracle_golden.py (CPU)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


# ---------------------------------------------------------------------------------- plane sweep ---------------

def _pixel_grid(h, w):
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    return torch.stack([xs + 0.5, ys + 0.5, torch.ones_like(xs)], 0).view(1, 3, h * w)   # geometry_utils.py:34-44


def _warp_plane(src, P, invK, pix, d, h, w):
    """One depth plane: world points, projected pixel coordinates, source-camera depth and the warped source features
    (reference cost_volume.py:139-234: BackprojectDepth, Project3D, grid_sample)."""
    b, k, c = src.shape[:3]
    pts = d.view(b, 1, -1) * torch.matmul(invK[:, :3, :3], pix)                       # [b,3,N]
    pts_bk = pts.repeat_interleave(k, 0)                                              # cost_volume.py:184-185
    q = torch.matmul(P[:, :3, :3], pts_bk) + P[:, :3, 3:4]                            # [b*k,3,N]
    z = q[:, 2:3] + 1e-8
    s = torch.where(q[:, 2:3].abs() > 1e-8, 1.0 / z, torch.ones_like(z))
    uv = q[:, :2] * s
    grid = (2.0 * uv * torch.tensor([1.0 / w, 1.0 / h]).view(1, 2, 1) - 1.0).view(b * k, 2, h, w).permute(0, 2, 3, 1)
    warped = F.grid_sample(src.reshape(b * k, c, h, w), grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = (z > 0).float().view(b, k, h, w)
    return pts, uv.view(b, k, 2, h, w), z.view(b, k, h, w), warped.view(b, k, c, h, w), mask


def dot_volume(cur, src, Ks, T_src_cur, invK, planes_bd):
    """CostVolumeManager.build_cost_volume + forward (cost_volume.py:237-380), plane by plane."""
    b, k, c, h, w = src.shape
    P = torch.matmul(Ks, T_src_cur).view(b * k, 4, 4)
    pix = _pixel_grid(h, w)
    vols = []
    for j in range(planes_bd.shape[1]):
        d = planes_bd[:, j].view(b, 1, 1).expand(b, 1, h * w)
        _, _, _, warped, mask = _warp_plane(src, P, invK, pix, d, h, w)
        vols.append(((warped * cur.unsqueeze(1)).sum(2) * mask).sum(1, keepdim=True))
    vol = torch.cat(vols, 1)
    lowest = torch.gather(planes_bd, 1, vol.argmax(1).view(b, -1)).view(b, h, w)
    return vol, lowest


def mlp_volume(cur, src, Ks, T_src_cur, T_cur_src, invK, planes_bd, lin):
    """FeatureVolumeManager.build_cost_volume (cost_volume.py:451-736): per plane the 202-channel (C(1+K)+10K+4) input
    tensor is materialised with torch.cat and pushed through the three Linear layers.  `lin` = [(W, b)] * 3."""
    b, k, c, h, w = src.shape
    n = h * w
    P = torch.matmul(Ks, T_src_cur).view(b * k, 4, 4)
    pix = _pixel_grid(h, w)
    # pose measures (geometry_utils.py:178-191)
    R, t = T_cur_src[:, :, :3, :3], T_cur_src[:, :, :3, 3]
    tr = R.diagonal(dim1=-2, dim2=-1).sum(-1)
    r_m = torch.sqrt(2.0 * (1.0 - torch.clamp(tr, max=3.0) / 3.0))
    t_m = t.norm(dim=-1)
    dist = torch.sqrt(t_m ** 2 + r_m ** 2)
    pose = [x.view(b, k, 1, 1).expand(b, k, h, w) for x in (dist, r_m, t_m)]
    vols, mask_any, bounds_any = [], None, None
    for j in range(planes_bd.shape[1]):
        d = planes_bd[:, j].view(b, 1, 1).expand(b, 1, n)
        pts, uv, z, warped, mask = _warp_plane(src, P, invK, pix, d, h, w)
        cur_ray = F.normalize(pts, dim=1).view(b, 3, h, w)                                        # :641-651
        src_ray = F.normalize(pts.unsqueeze(1) - t.unsqueeze(-1), dim=2).view(b, k, 3, h, w)        # :654-669
        angle = F.cosine_similarity(cur_ray.unsqueeze(1).expand(b, k, 3, h, w), src_ray, dim=2, eps=1e-5)  # :683-688
        dots = (warped * cur.unsqueeze(1)).sum(2) * mask                                             # :691-695
        x = torch.cat([warped.reshape(b, k * c, h, w), cur, mask, z, d.view(b, 1, h, w), dots, angle, cur_ray,
                       src_ray.reshape(b, 3 * k, h, w)] + pose, 1)                                   # :709-723
        x = x.permute(0, 2, 3, 1)
        x = F.leaky_relu(F.linear(x, *lin[0]), 0.01)
        x = F.leaky_relu(F.linear(x, *lin[1]), 0.01)
        vols.append(F.linear(x, *lin[2]).permute(0, 3, 1, 2))
        if j == planes_bd.shape[1] - 1:                                                             # :544-551, 625-637
            inb = (uv[:, :, 0] > 2) & (uv[:, :, 0] < w - 2) & (uv[:, :, 1] > 2) & (uv[:, :, 1] < h - 2)
            mask_any, bounds_any = (mask > 0).any(1), inb.any(1)
    vol = torch.cat(vols, 1)
    lowest = torch.gather(planes_bd, 1, vol.argmax(1).view(b, -1)).view(b, h, w)
    return vol, lowest, mask_any & bounds_any


# ---------------------------------------------------------------------------------- conv stack -----------------

def _block(x, m):
    """BasicBlock with norm_layer = Identity (layers.py:68-85) on the weights of holder module `m`."""
    out = F.leaky_relu(F.conv2d(x, m.conv1.weight, m.conv1.bias, stride=m.conv1.stride, padding=1), 0.2)
    out = F.conv2d(out, m.conv2.weight, m.conv2.bias, padding=1)
    if m.downsample is not None:
        ds = m.downsample[0]
        x = F.conv2d(x, ds.weight, ds.bias, stride=ds.stride, padding=ds.padding)
    return F.leaky_relu(out + x, 0.2)


def _seq(x, mods):
    for m in mods:
        x = _block(x, m)
    return x


def _up(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)   # generic_utils.py:96-105


def cv_encoder(net, vol, img_feats):
    outs, x = [], vol
    for i in range(net.num_blocks):
        x = _block(x, net.convs[f"ds_conv_{i}"])
        x = _seq(torch.cat([x, img_feats[i]], 1), net.convs[f"conv_{i}"])
        outs.append(x)
    return outs


def depth_decoder(net, feats):
    """UNet++ (networks.py:75-96), every head evaluation included as the reference does."""
    prev, outs, res = list(feats), [], {}
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            parts = [_block(prev[i], net.convs[f"right_conv_{i}{j - 1}"]),
                     _up(_block(prev[i + 1], net.convs[f"diag_conv_{i + 1}{j - 1}"]))]
            if i + j != 4:
                parts.append(_up(_block(outs[-1], net.convs[f"up_conv_{i + 1}{j}"])))
            node = _seq(torch.cat(parts, 1), net.convs[f"in_conv_{i}{j}"])
            outs.append(node)
            head = net.convs[f"output_{i}"]
            y = node if isinstance(head[0], torch.nn.Identity) else _block(node, head[0])
            res[f"log_depth_pred_s{i}_b1hw"] = F.conv2d(y, head[1].weight, head[1].bias)
        prev = outs[::-1]
    return res


# ---------------------------------------------------------------------------------- encoders -------------------

def matching_encoder(net, images):
    """ResnetMatchingEncoder (networks.py:176-201) with the antialiased ResNet-18 stem restated in ATen ops
    (oracle/refshim.py documents the restatement; the package is absent)."""
    sd = net.state_dict()

    def bn(x, pre):
        return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                            training=False, eps=1e-5)
    x = F.relu(bn(F.conv2d(images, sd["net.0.weight"], None, stride=2, padding=3), "net.1."))
    x = F.max_pool2d(x, 2, stride=1)
    x = F.conv2d(F.pad(x, (1, 2, 1, 2), mode="reflect"), sd["net.3.1.filt"], stride=2, groups=64)
    for blk in ("net.4.0.", "net.4.1."):
        y = F.relu(bn(F.conv2d(x, sd[blk + "conv1.weight"], None, padding=1), blk + "bn1."))
        y = bn(F.conv2d(y, sd[blk + "conv2.weight"], None, padding=1), blk + "bn2.")
        x = F.relu(y + x)
    x = F.conv2d(x, sd["net.5.weight"], sd["net.5.bias"])
    x = F.leaky_relu(F.instance_norm(x), 0.2)
    x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd["net.8.weight"], sd["net.8.bias"])
    return F.instance_norm(x)


def image_prior_encoder(net, image):
    import effnet_torch   # tests/effnet_torch.py: the ATen statement of tf_efficientnetv2_s (features_only)
    return effnet_torch.features(image, net.state_dict())


# ---------------------------------------------------------------------------------- whole step -----------------

def hero_forward(model, cur_image, src_image, T_src_cur, T_cur_src, Ks, invK, min_depth, max_depth, planes_bd,
                 with_prior=True, with_encoder=True, pyramid=None, feats=None, mlp=True):
    """DepthModel.forward_tensors on CPU for one batch (reference depth_model.py:358-405)."""
    b, k = src_image.shape[:2] if src_image is not None else feats[1].shape[:2]
    if with_prior:
        pyramid = image_prior_encoder(model.encoder, cur_image)
    if with_encoder:
        allf = matching_encoder(model.matching_model,
                                torch.cat([cur_image.unsqueeze(1), src_image], 1).flatten(0, 1)).unflatten(0, (b, k + 1))
        cur_f, src_f = allf[:, 0], allf[:, 1:]
    else:
        cur_f, src_f = feats
    if mlp:
        lin = [(m.weight, m.bias) for m in model.cost_volume.mlp.net if isinstance(m, torch.nn.Linear)]
        vol, lowest, mask = mlp_volume(cur_f, src_f, Ks, T_src_cur, T_cur_src, invK, planes_bd, lin)
    else:
        vol, lowest = dot_volume(cur_f, src_f, Ks, T_src_cur, invK, planes_bd)
    enc = cv_encoder(model.cost_volume_net, vol, pyramid[1:])
    out = depth_decoder(model.depth_decoder, [pyramid[0]] + enc)
    out["depth_pred_s0_b1hw"] = torch.exp(out["log_depth_pred_s0_b1hw"])
    out["lowest_cost_bhw"] = lowest
    return out
