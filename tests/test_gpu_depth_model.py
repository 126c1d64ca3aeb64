"""End-to-end hot path (DepthModel.hot_path: sweep -> CVEncoder -> DepthDecoderPP -> exp) on the GPU
against the same chain through the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle
from parity import elementwise_rel_percentiles, assert_close, mismatch_fraction
from simplerecon_amd import depth_model as dm
from simplerecon_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("fvt", ["mlp_feature_volume", "simple_cost_volume"])
def test_hot_path_matches_oracle_chain(fvt):
    B, K, C, D, h, w = 2, 3, 16, 8, 24, 32
    opts = dm.default_options(image_width=4 * w, image_height=4 * h, model_num_views=K + 1,
                              matching_num_depth_bins=D, feature_volume_type=fvt)
    model = dm.DepthModel(opts)
    synthetic.seeded_fill_(model.cost_volume_net, seed=1)
    synthetic.seeded_fill_(model.depth_decoder, seed=2)
    if fvt == "mlp_feature_volume":
        synthetic.seeded_fill_(model.cost_volume.mlp, seed=3)
    model = model.to(DEV).eval()
    inp = synthetic.cost_volume_inputs(B, K, C, h, w, seed=5)
    pyr = synthetic.image_prior_pyramid(B, h, w, seed=5)
    d = {k: v.to(DEV) for k, v in inp.items()}
    with torch.inference_mode():
        out = model.hot_path([f.to(DEV) for f in pyr], d["cur_feats"], d["src_feats"], d["src_extrinsics"],
                             d["src_poses"], d["src_Ks"], d["cur_invK"], return_mask=True)
    torch.cuda.synchronize()
    n = {k: v.numpy() for k, v in inp.items()}
    planes = model.cost_volume.generate_depth_planes(B, d["min_depth"], d["max_depth"])[:, :, 0, 0].cpu().numpy()
    if fvt == "mlp_feature_volume":
        sd = {k: v.cpu().numpy() for k, v in model.cost_volume.mlp.state_dict().items()}
        mlp = dict(W1=sd["net.0.weight"], b1=sd["net.0.bias"], W2=sd["net.2.weight"], b2=sd["net.2.bias"],
                   W3=sd["net.4.weight"], b3=sd["net.4.bias"])
        vol, low, mask = oracle.mlp_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"],
                                           n["src_poses"], n["cur_invK"], planes, mlp, want_mask=True)
        assert mismatch_fraction(out["overall_mask_bhw"], mask) == 0.0
    else:
        vol, low, _ = oracle.dot_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"],
                                        n["cur_invK"], planes)
        assert out["overall_mask_bhw"] is None
    esd = {k: v.cpu().numpy() for k, v in model.cost_volume_net.state_dict().items()}
    dsd = {k: v.cpu().numpy() for k, v in model.depth_decoder.state_dict().items()}
    feats = oracle.cv_encoder(vol, [f.numpy() for f in pyr[1:]], esd)
    ref = oracle.depth_decoder_pp([pyr[0].numpy()] + feats, dsd)
    for i in range(4):
        k = f"log_depth_pred_s{i}_b1hw"
        assert_close(out[k], ref[k], what=k)
        assert_close(out[k.replace("log_", "")], np.exp(ref[k]), what="depth " + k)
    assert out["depth_pred_s0_b1hw"].shape == (B, 1, 2 * h, 2 * w)
    assert out["lowest_cost_bhw"].shape == (B, h, w)


def test_forward_all_native_matches_oracle_chain():
    """DepthModel.forward with BOTH encoders native (default construction) against the whole chain through the CPU
    oracle: EfficientNetV2-S pyramid, ResnetMatchingEncoder on cur + source images, metadata-MLP sweep, CVEncoder,
    DepthDecoderPP, exp."""
    B, K, H, W, D = 1, 2, 96, 128, 8
    h, w = H // 4, W // 4
    opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=D)
    model = dm.DepthModel(opts)
    assert type(model.encoder).__name__ == "EfficientNetV2SFeatures"
    assert type(model.matching_model).__name__ == "ResnetMatchingEncoder"
    synthetic.seeded_fill_(model.encoder, seed=6, gain=1.0)
    for i, m in enumerate((model.matching_model, model.cost_volume_net, model.depth_decoder, model.cost_volume.mlp)):
        synthetic.seeded_fill_(m, seed=20 + i)
    model = model.to(DEV).eval()
    inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=4)
    g = torch.Generator().manual_seed(8)
    cur_img, src_img = torch.randn((B, 3, H, W), generator=g), torch.randn((B, K, 3, H, W), generator=g)
    eye = torch.eye(4).expand(B, 4, 4).contiguous()
    cur = {"image_b3hw": cur_img.to(DEV), "invK_s1_b44": inp["cur_invK"].to(DEV), "cam_T_world_b44": eye.to(DEV),
           "world_T_cam_b44": eye.to(DEV)}
    src = {"image_b3hw": src_img.to(DEV), "K_s1_b44": inp["src_Ks"].to(DEV),
           "cam_T_world_b44": inp["src_extrinsics"].to(DEV), "world_T_cam_b44": inp["src_poses"].to(DEV)}
    with torch.inference_mode():
        out = model("test", cur, src, return_mask=True)
        unb = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
    torch.cuda.synchronize()

    def sd(m):
        return {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    n = {k: v.numpy() for k, v in inp.items()}
    pyr = oracle.efficientnetv2_s_features(cur_img.numpy(), sd(model.encoder))
    msd = sd(model.matching_model)
    mcur = oracle.resnet_matching_encoder(cur_img.numpy(), msd)
    msrc = oracle.resnet_matching_encoder(src_img.numpy().reshape(B * K, 3, H, W), msd).reshape(B, K, 16, h, w)
    planes = model.cost_volume.generate_depth_planes(B, inp["min_depth"].to(DEV), inp["max_depth"].to(DEV))
    planes = planes[:, :, 0, 0].cpu().numpy()
    ms = sd(model.cost_volume.mlp)
    mlp = dict(W1=ms["net.0.weight"], b1=ms["net.0.bias"], W2=ms["net.2.weight"], b2=ms["net.2.bias"],
               W3=ms["net.4.weight"], b3=ms["net.4.bias"])
    vol, low, mask = oracle.mlp_volume(mcur, msrc, n["src_Ks"], n["src_extrinsics"], n["src_poses"], n["cur_invK"],
                                       planes, mlp, want_mask=True)
    feats = oracle.cv_encoder(vol, pyr[1:], sd(model.cost_volume_net))
    ref = oracle.depth_decoder_pp([pyr[0]] + feats, sd(model.depth_decoder))
    assert mismatch_fraction(out["overall_mask_bhw"], mask) == 0.0
    for i in range(4):
        k = f"log_depth_pred_s{i}_b1hw"
        assert_close(out[k], ref[k], what=k)
        assert_close(out[k.replace("log_", "")], np.exp(ref[k]), what="depth " + k)
        assert_close(unb[k], ref[k], what=k + " (unbatched matching encoder)")
    # element-wise (not range-relative) error of the metric depth map: every element has a meaningful scale
    pct = elementwise_rel_percentiles(out["depth_pred_s0_b1hw"], np.exp(ref["log_depth_pred_s0_b1hw"]))
    assert pct["p99"] < 1e-4 and pct["max"] < 1e-3, pct


def test_forward_api_signature_and_output_keys():
    """DepthModel.forward keeps the reference's call signature and output keys (depth_model.py:247-407)."""
    B, K, H, W = 1, 2, 96, 128
    opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=8)
    model = dm.DepthModel(opts)
    model = model.to(DEV).eval()
    inp = synthetic.cost_volume_inputs(B, K, 16, H // 4, W // 4, seed=2, device=DEV)
    eye = torch.eye(4, device=DEV)
    cur = {"image_b3hw": torch.randn(B, 3, H, W, device=DEV), "invK_s1_b44": inp["cur_invK"],
           "cam_T_world_b44": eye.expand(B, 4, 4).contiguous(), "world_T_cam_b44": eye.expand(B, 4, 4).contiguous()}
    src = {"image_b3hw": torch.randn(B, K, 3, H, W, device=DEV), "K_s1_b44": inp["src_Ks"],
           "cam_T_world_b44": inp["src_extrinsics"], "world_T_cam_b44": inp["src_poses"]}
    with torch.inference_mode():
        out = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
    keys = set(out)
    assert {f"log_depth_pred_s{i}_b1hw" for i in range(4)} <= keys and {f"depth_pred_s{i}_b1hw" for i in range(4)} <= keys
    assert {"lowest_cost_bhw", "overall_mask_bhw"} <= keys
    assert out["depth_pred_s0_b1hw"].shape == (B, 1, H // 2, W // 2) and torch.isfinite(out["depth_pred_s0_b1hw"]).all()
    # under autograd the same call is the training forward (every stage differentiable on HIP kernels): outputs carry a graph
    with torch.enable_grad():
        pyr = [t.detach().requires_grad_() for t in model.encoder(cur["image_b3hw"])]
        tr = model.hot_path(pyr, inp["cur_feats"], inp["src_feats"].requires_grad_(), inp["src_extrinsics"], inp["src_poses"],
                            inp["src_Ks"], inp["cur_invK"])
    assert tr["depth_pred_s0_b1hw"].requires_grad and not tr["lowest_cost_bhw"].requires_grad
    # geometry is data, as in the reference: a pose that asks for a gradient is refused
    with pytest.raises(NotImplementedError):
        with torch.enable_grad():
            model.hot_path(pyr, inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"].clone().requires_grad_(),
                           inp["src_poses"], inp["src_Ks"], inp["cur_invK"])


def test_full_size_hot_path_properties():
    """BASELINE.json configs[2] shapes (640x480, 7 views, 64 planes), batch 2: size-independent properties --
    finite outputs, run-to-run determinism, independence of the frames of a batch, positive depths that are
    exp(log-depth), mask/argmax consistency of the sweep outputs."""
    B, K, C, D, h, w = 2, 7, 16, 64, 120, 160
    opts = dm.default_options(image_width=4 * w, image_height=4 * h, model_num_views=K + 1, matching_num_depth_bins=D)
    model = dm.DepthModel(opts)
    synthetic.seeded_fill_(model.cost_volume_net, seed=1)
    synthetic.seeded_fill_(model.depth_decoder, seed=2)
    synthetic.seeded_fill_(model.cost_volume.mlp, seed=3)
    model = model.to(DEV).eval()
    inp = synthetic.cost_volume_inputs(B, K, C, h, w, seed=11, device=DEV)
    pyr = synthetic.image_prior_pyramid(B, h, w, seed=11, device=DEV)

    def run(sl=slice(None)):
        with torch.inference_mode():
            return model.hot_path([f[sl] for f in pyr], inp["cur_feats"][sl], inp["src_feats"][sl],
                                  inp["src_extrinsics"][sl], inp["src_poses"][sl], inp["src_Ks"][sl],
                                  inp["cur_invK"][sl], return_mask=True)
    # (six runs: at batch 2 the decoder's branches run on side HIP streams, and a kernel whose result depends on how its waves
    # interleave -- r05: the first wave-specialised Winograd form -- differs in roughly two runs out of five, not in every one)
    a, b, *more = [run() for _ in range(6)]
    one = run(slice(1, 2))
    for i in range(4):
        k = f"depth_pred_s{i}_b1hw"
        assert a[k].shape == (B, 1, (2 * h) >> i, (2 * w) >> i)
        assert torch.isfinite(a[k]).all() and bool((a[k] > 0).all())
        assert torch.equal(a[k], b[k]) and all(torch.equal(a[k], m[k]) for m in more), "hot path is not deterministic"
        assert torch.equal(a[k], torch.exp(a[k.replace("depth_", "log_depth_")]))
        # the cost volume is bitwise batch-independent (tests/test_gpu_mlp_volume.py); in the conv stack the launch
        # plan (split-K of the deep layers) depends on the batch size, hence the summation order: agreement, not identity
        assert_close(one[k][0], a[k][1], tol=1e-5, what=f"frame 1 alone vs inside the batch, {k}")
    assert torch.equal(a["lowest_cost_bhw"], b["lowest_cost_bhw"]) and a["overall_mask_bhw"].dtype == torch.bool
    assert 0.5 < float(a["overall_mask_bhw"].float().mean()) <= 1.0
