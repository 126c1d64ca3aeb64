"""BASELINE.json configs[4], the stress configuration, END TO END: 15 source views, 96 depth planes, batch 1, at 960x736.

The reference itself cannot run its UNet++ at 960x720 (modules/networks.py:83-89: the pyramid is 360/180/90/45/23 rows and
`upsample(23) = 46 != 45`; reproduced in SURVEY.md §7), which is why r01-r03 covered this configuration with the cost volume
only (`hero_cfg5_volume`).  SURVEY §7's variant -- pad the image height to 736 = 23 * 32 -- makes every level even; this
test runs `DepthModel.forward_tensors` there with the production dispatch and compares it with the whole CPU oracle chain
(EfficientNetV2-S pyramid, matching encoder on 1 + 15 images, 410-input metadata-MLP sweep over 96 planes at 184x240,
CVEncoder, DepthDecoderPP, exp): the Winograd / direct / pointwise launch plans at 368x480, 184x240, 92x120, 46x60 and
23x30 maps (odd row counts at the deepest level, ragged 8x16 regions everywhere) that no 640x480 test reaches."""
import numpy as np
import pytest
import torch

import oracle
from parity import assert_close, assert_lowest_cost, elementwise_rel_percentiles, mismatch_fraction
from simplerecon_amd import depth_model as dm
from simplerecon_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, K, D, H, W = 1, 15, 96, 736, 960
h, w = H // 4, W // 4


def _sd(m):
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def test_stress_config_end_to_end_at_960x736():
    opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=D)
    model = dm.DepthModel(opts)
    for name, seed in (("matching_model", 4), ("encoder", 5), ("cost_volume_net", 1), ("depth_decoder", 2)):
        synthetic.seeded_fill_(getattr(model, name), seed=seed)
    synthetic.seeded_fill_(model.cost_volume.mlp, seed=3)
    model = model.to(DEV).eval()
    assert model.cost_volume.mlp.net[0].in_features == 16 * (K + 1) + 10 * K + 4 == 410
    inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=0)
    g = torch.Generator(device="cpu").manual_seed(77)
    cur, src = torch.randn((B, 3, H, W), generator=g), torch.randn((B, K, 3, H, W), generator=g)
    got = {}
    hooks = [model.cost_volume.register_forward_hook(lambda m, a, o: got.__setitem__("cv", o)),
             model.cost_volume_net.register_forward_hook(lambda m, a, o: got.__setitem__("levels", o))]
    d = {k: v.to(DEV) for k, v in inp.items() if k not in ("min_depth", "max_depth")}
    with torch.inference_mode():
        out = model.forward_tensors(cur.to(DEV), src.to(DEV), d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                                    return_mask=True)
    torch.cuda.synchronize()
    for hk in hooks:
        hk.remove()
    assert out["depth_pred_s0_b1hw"].shape == (B, 1, H // 2, W // 2)
    assert [tuple(f.shape[2:]) for f in got["levels"]] == [(184, 240), (92, 120), (46, 60), (23, 30)]

    # ---- the oracle chain ----
    n = {k: v.numpy() for k, v in inp.items() if k not in ("min_depth", "max_depth")}
    pyr = oracle.efficientnetv2_s_features(cur.numpy(), _sd(model.encoder))
    msd = _sd(model.matching_model)
    mcur = oracle.resnet_matching_encoder(cur.numpy(), msd)
    msrc = oracle.resnet_matching_encoder(src[0].numpy(), msd).reshape(1, K, 16, h, w)
    planes = model.cost_volume.generate_depth_planes(1, inp["min_depth"].to(DEV), inp["max_depth"].to(DEV))[:, :, 0, 0].cpu().numpy()
    ms = _sd(model.cost_volume.mlp)
    mlp = dict(W1=ms["net.0.weight"], b1=ms["net.0.bias"], W2=ms["net.2.weight"], b2=ms["net.2.bias"],
               W3=ms["net.4.weight"], b3=ms["net.4.bias"])
    vol, low, mask = oracle.mlp_volume(mcur, msrc, n["src_Ks"], n["src_extrinsics"], n["src_poses"], n["cur_invK"], planes, mlp,
                                       want_mask=True)
    levels = oracle.cv_encoder(vol, pyr[1:], _sd(model.cost_volume_net))
    ref = oracle.depth_decoder_pp([pyr[0]] + levels, _sd(model.depth_decoder))

    assert_close(got["cv"][0], vol, what="cost volume")
    assert mismatch_fraction(out["overall_mask_bhw"], mask) == 0.0
    assert_lowest_cost(out["lowest_cost_bhw"], got["cv"][0], planes, low, what="stress config")
    for lv, (a, b) in enumerate(zip(got["levels"], levels)):
        assert_close(a, b, what=f"CVEncoder level {lv}")
    for s in range(4):
        k = f"log_depth_pred_s{s}_b1hw"
        assert out[k].shape[1:] == (1, (H // 2) >> s, (W // 2) >> s)
        assert_close(out[k], ref[k], what=k)
    pct = elementwise_rel_percentiles(out["depth_pred_s0_b1hw"], np.exp(ref["log_depth_pred_s0_b1hw"]))
    assert pct["p99"] < 3e-5 and pct["max"] < 1e-4, pct
