"""BASELINE.json configs[4], the stress configuration, END TO END: 15 source views, 96 depth planes, batch 1, at 960x736.

The reference itself cannot run its UNet++ at 960x720 (modules/networks.py:83-89: the pyramid is 360/180/90/45/23 rows and
`upsample(23) = 46 != 45`; reproduced in SURVEY.md §7), which is why r01-r03 covered this configuration with the cost volume
only (`hero_cfg5_volume`).  SURVEY §7's variant -- pad the image height to 736 = 23 * 32 -- makes every level even; this
test runs `DepthModel.forward_tensors` there with the production dispatch and compares it with the whole CPU oracle chain
(EfficientNetV2-S pyramid, matching encoder on 1 + 15 images, 410-input metadata-MLP sweep over 96 planes at 184x240,
CVEncoder, DepthDecoderPP, exp): the Winograd / direct / pointwise launch plans at 368x480, 184x240, 92x120, 46x60 and
23x30 maps (odd row counts at the deepest level, ragged 8x16 regions everywhere) that no 640x480 test reaches.

r05: the bench workload `hero_cfg5` runs this configuration at BATCH 4, whose launch plans (work items
per layer, split-K factors, F(4x4) / F(2x2) Winograd choice, pointwise tiles) differ from batch 1's: the second test runs the
batch-4 forward and checks frames 0 and 3 against the same oracle chain."""
import numpy as np
import pytest
import torch

import oracle
from parity import assert_close, assert_lowest_cost, elementwise_rel_percentiles, mismatch_fraction, capture_cv_encoder_levels
from simplerecon_amd import depth_model as dm
from simplerecon_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BMAX, K, D, H, W = 4, 15, 96, 736, 960
h, w = H // 4, W // 4


def _sd(m):
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


class _Case:
    """Model + four seeded keyframes + the per-frame oracle chain (computed once per frame, cached)."""

    def __init__(self):
        opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=D)
        model = dm.DepthModel(opts)
        for name, seed in (("matching_model", 4), ("encoder", 5), ("cost_volume_net", 1), ("depth_decoder", 2)):
            synthetic.seeded_fill_(getattr(model, name), seed=seed)
        synthetic.seeded_fill_(model.cost_volume.mlp, seed=3)
        self.model = model.to(DEV).eval()
        assert model.cost_volume.mlp.net[0].in_features == 16 * (K + 1) + 10 * K + 4 == 410
        one = synthetic.cost_volume_inputs(1, K, 16, h, w, seed=0)          # frame 0 = the r04 test's keyframe
        rest = synthetic.cost_volume_inputs(BMAX - 1, K, 16, h, w, seed=21)  # frames 1-3: other poses
        self.inp = {k: (torch.cat([one[k], rest[k]], 0) if one[k].dim() > 0 and one[k].shape[0] == 1 and
                        k not in ("min_depth", "max_depth") else one[k]) for k in one}
        g = torch.Generator(device="cpu").manual_seed(77)
        cur0, src0 = torch.randn((1, 3, H, W), generator=g), torch.randn((1, K, 3, H, W), generator=g)
        g = torch.Generator(device="cpu").manual_seed(78)
        self.cur = torch.cat([cur0, torch.randn((BMAX - 1, 3, H, W), generator=g)], 0)
        self.src = torch.cat([src0, torch.randn((BMAX - 1, K, 3, H, W), generator=g)], 0)
        self._oracle = {}

    def run_hip(self, sl):
        got = {}
        m = self.model
        hooks = [m.cost_volume.register_forward_hook(lambda mod, a, o: got.__setitem__("cv", o)),
                 m.cost_volume_net.register_forward_hook(capture_cv_encoder_levels(got))]
        d = {k: v[sl].to(DEV) for k, v in self.inp.items() if k not in ("min_depth", "max_depth")}
        with torch.inference_mode():
            out = m.forward_tensors(self.cur[sl].to(DEV), self.src[sl].to(DEV), d["src_extrinsics"], d["src_poses"], d["src_Ks"],
                                    d["cur_invK"], return_mask=True)
        torch.cuda.synchronize()
        for hk in hooks:
            hk.remove()
        return out, got

    def oracle(self, f):
        if f not in self._oracle:
            m, inp = self.model, self.inp
            n = {k: v[f:f + 1].numpy() for k, v in inp.items() if k not in ("min_depth", "max_depth")}
            pyr = oracle.efficientnetv2_s_features(self.cur[f:f + 1].numpy(), _sd(m.encoder))
            msd = _sd(m.matching_model)
            mcur = oracle.resnet_matching_encoder(self.cur[f:f + 1].numpy(), msd)
            msrc = oracle.resnet_matching_encoder(self.src[f].numpy(), msd).reshape(1, K, 16, h, w)
            planes = m.cost_volume.generate_depth_planes(1, inp["min_depth"].to(DEV), inp["max_depth"].to(DEV))[:, :, 0, 0].cpu().numpy()
            ms = _sd(m.cost_volume.mlp)
            mlp = dict(W1=ms["net.0.weight"], b1=ms["net.0.bias"], W2=ms["net.2.weight"], b2=ms["net.2.bias"],
                       W3=ms["net.4.weight"], b3=ms["net.4.bias"])
            vol, low, mask = oracle.mlp_volume(mcur, msrc, n["src_Ks"], n["src_extrinsics"], n["src_poses"], n["cur_invK"], planes,
                                               mlp, want_mask=True)
            levels = oracle.cv_encoder(vol, pyr[1:], _sd(m.cost_volume_net))
            ref = oracle.depth_decoder_pp([pyr[0]] + levels, _sd(m.depth_decoder))
            self._oracle[f] = dict(vol=vol, low=low, mask=mask, levels=levels, ref=ref, planes=planes)
        return self._oracle[f]

    def check(self, out, got, b, f, what):
        """frame `b` of the HIP outputs against the oracle chain of keyframe `f`"""
        o = self.oracle(f)
        sl = slice(b, b + 1)
        assert_close(got["cv"][0][sl], o["vol"], what=f"{what}: cost volume")
        assert mismatch_fraction(out["overall_mask_bhw"][sl], o["mask"]) == 0.0
        assert_lowest_cost(out["lowest_cost_bhw"][sl], got["cv"][0][sl], o["planes"], o["low"], what=what)
        for lv, (a, r) in enumerate(zip(got["levels"], o["levels"])):
            assert_close(a[sl], r, what=f"{what}: CVEncoder level {lv}")
        for s in range(4):
            k = f"log_depth_pred_s{s}_b1hw"
            assert out[k].shape[1:] == (1, (H // 2) >> s, (W // 2) >> s)
            assert_close(out[k][sl], o["ref"][k], what=f"{what}: {k}")
        pct = elementwise_rel_percentiles(out["depth_pred_s0_b1hw"][sl], np.exp(o["ref"]["log_depth_pred_s0_b1hw"]))
        assert pct["p99"] < 3e-5 and pct["max"] < 1e-4, (what, pct)


@pytest.fixture(scope="module")
def case():
    return _Case()


def test_stress_config_end_to_end_at_960x736(case):
    out, got = case.run_hip(slice(0, 1))
    assert out["depth_pred_s0_b1hw"].shape == (1, 1, H // 2, W // 2)
    assert [tuple(f.shape[2:]) for f in got["levels"]] == [(184, 240), (92, 120), (46, 60), (23, 30)]
    case.check(out, got, 0, 0, "batch 1, frame 0")


def test_stress_config_at_the_benchmarked_batch_4(case):
    """`hero_cfg5` is timed at batch 4: the plans of THAT batch (2 760 F(4x4) work items per full-resolution layer instead of
    690, other split-K factors below) on frames 0 and 3."""
    out, got = case.run_hip(slice(0, BMAX))
    assert out["depth_pred_s0_b1hw"].shape == (BMAX, 1, H // 2, W // 2)
    case.check(out, got, 0, 0, "batch 4, frame 0")
    case.check(out, got, 3, 3, "batch 4, frame 3")
