"""GPU parity of the fused metadata-MLP sweep (sr_mlp_volume_fwd via FeatureVolumeManager /
FastFeatureVolumeManager / to_hip) against the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle
from parity import assert_close, assert_lowest_cost, mismatch_fraction, rel_err
from simplerecon_amd import synthetic
from simplerecon_amd.cost_volume import FastFeatureVolumeManager, FeatureVolumeManager

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _manager(case, cls=FeatureVolumeManager):
    mgr = cls(case["h"], case["w"], num_depth_bins=case["D"], matching_dim_size=case["C"],
              num_source_views=case["K"])
    synthetic.seeded_fill_(mgr.mlp, seed=case["seed"])
    return mgr.to(DEV)


def _run(mgr, inp, return_mask=True):
    dinp = {k: v.to(DEV) for k, v in inp.items()}
    with torch.inference_mode():
        out = mgr(return_mask=return_mask, **dinp)
    torch.cuda.synchronize()
    return out


def _mlp_np(mgr):
    sd = {k: v.cpu().numpy() for k, v in mgr.mlp.state_dict().items()}
    return dict(W1=sd["net.0.weight"], b1=sd["net.0.bias"], W2=sd["net.2.weight"], b2=sd["net.2.bias"],
                W3=sd["net.4.weight"], b3=sd["net.4.bias"])


def _oracle(mgr, inp, planes_np, precision="f32"):
    n = {k: v.numpy() for k, v in inp.items()}
    return oracle.mlp_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["src_poses"],
                             n["cur_invK"], planes_np, _mlp_np(mgr), want_mask=True, precision=precision)


@pytest.mark.parametrize("name", [n for n, c in gc.VOLUME_CASES.items() if c["model"] == "hero"])
def test_mlp_volume_matches_oracle_and_golden(name):
    case = gc.VOLUME_CASES[name]
    inp = gc.volume_inputs(case)
    gold = gc.load_golden("volume", name)
    mgr = _manager(case)
    vol, lowest, planes, mask = _run(mgr, inp)
    planes_np = planes.cpu().numpy() if "depth_planes_bdhw" in inp else planes[:, :, 0, 0].cpu().numpy()
    cv_o, low_o, mask_o = _oracle(mgr, inp, planes_np)
    assert_close(vol, cv_o, tol=2e-5, what=f"{name} vs oracle")
    assert_close(vol, gold["cost_volume"], what=f"{name} vs reference golden (FeatureVolumeManager)")
    assert_close(vol, gold["cost_volume_fast"], what=f"{name} vs reference golden (FastFeatureVolumeManager)")
    assert mask.dtype == torch.bool
    assert mismatch_fraction(mask, mask_o) == 0.0 and mismatch_fraction(mask, gold["overall_mask"]) == 0.0
    assert_lowest_cost(lowest, vol, planes_np, gold["lowest_cost"], name)
    # fp32 MFMA result is as close to the fp64 truth as the reference's own fp32 result is
    cv64, _, _ = _oracle(mgr, inp, planes_np, "f64")
    assert rel_err(vol, cv64) < 2 * max(rel_err(gold["cost_volume"], cv64), 1e-6)


def test_to_fast_and_no_mask():
    case = gc.VOLUME_CASES["hero_small"]
    inp = gc.volume_inputs(case)
    mgr = _manager(case)
    fast = mgr.to_fast()
    assert isinstance(fast, FastFeatureVolumeManager) and fast.mlp is mgr.mlp
    a = _run(mgr, inp)
    b = _run(fast, inp, return_mask=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and b[3] is None


def test_state_dict_round_trip_with_reference_names():
    case = gc.VOLUME_CASES["hero_k7"]
    mgr = _manager(case)
    keys = list(mgr.state_dict())
    assert keys == ["linear_ramp_1d11", "backprojector.pix_coords_13N", "projector.eps", "mlp.net.0.weight",
                    "mlp.net.0.bias", "mlp.net.2.weight", "mlp.net.2.bias", "mlp.net.4.weight", "mlp.net.4.bias"]
    other = FeatureVolumeManager(case["h"], case["w"], num_depth_bins=case["D"]).to(DEV)
    other.load_state_dict(mgr.state_dict(), strict=True)
    inp = gc.volume_inputs(case)
    assert torch.equal(_run(mgr, inp)[0], _run(other, inp)[0])


@pytest.mark.parametrize("B", [1, 2])
def test_cfg3_shape_full_resolution(B):
    """hero_model.yaml shapes (7 views, 64 planes, 120x160, 202-channel MLP) at batch 1 and 2,
    checked against the oracle on a strided subset of planes (full oracle would take minutes)."""
    case = dict(B=B, K=7, C=16, D=64, h=120, w=160, seed=50 + B)
    inp = synthetic.cost_volume_inputs(B, 7, 16, 120, 160, seed=case["seed"])
    mgr = _manager(case)
    vol, lowest, planes, mask = _run(mgr, inp)
    planes_np = planes[:, :, 0, 0].cpu().numpy()
    sub = [0, 21, 42, 63]
    cv_o, _, mask_o = _oracle(mgr, inp, planes_np[:, sub])
    assert_close(vol[:, sub], cv_o, tol=2e-5, what="cfg3 planes subset vs oracle")
    assert mismatch_fraction(mask, mask_o) == 0.0  # plane 63 is the last of both
    assert_lowest_cost(lowest, vol, planes_np, lowest.cpu().numpy(), "cfg3 self-consistency")
    # channels-last volume (what the HIP CVEncoder consumes) is the same volume
    mgr.volume_memory_format = torch.channels_last
    vol_cl = _run(mgr, inp)[0]
    assert vol_cl.is_contiguous(memory_format=torch.channels_last) and torch.equal(vol_cl.contiguous(), vol)
    # frames of a batch are independent
    if B > 1:
        one = {k: (v[1:2].contiguous() if v.dim() > 0 and v.shape[0] == B else v) for k, v in inp.items()}
        mgr.volume_memory_format = torch.contiguous_format
        v1 = _run(mgr, one)[0]
        assert torch.equal(v1[0], vol[1])


def test_unsupported_configs_fail_loudly():
    from simplerecon_amd._lib import HipLibraryError
    case = dict(B=1, K=2, C=8, D=4, h=8, w=12, seed=1)
    mgr = _manager(case)
    inp = synthetic.cost_volume_inputs(1, 2, 8, 8, 12)
    with pytest.raises(HipLibraryError):
        _run(mgr, inp)


@pytest.mark.parametrize("K", [1, 9, 12, 15])
def test_mlp_volume_other_view_counts(K):
    """Source-view counts that exercise the other weight placements of the sweep kernel: K <= 7 keeps W2 and the
    depth-variant part of W1 in LDS, K = 8..12 streams W2 from L2, K >= 13 streams W1 (BASELINE configs[4]: K = 15)."""
    case = dict(B=1, K=K, C=16, D=5, h=12, w=20, seed=70 + K)
    inp = synthetic.cost_volume_inputs(case["B"], K, 16, case["h"], case["w"], seed=case["seed"])
    mgr = _manager(case)
    vol, lowest, planes, mask = _run(mgr, inp)
    planes_np = planes[:, :, 0, 0].cpu().numpy()
    cv_o, low_o, mask_o = _oracle(mgr, inp, planes_np)
    assert_close(vol, cv_o, tol=2e-5, what=f"K={K} vs oracle")
    assert mismatch_fraction(mask, mask_o) == 0.0


def test_cfg3_batch8_full_size_all_frames():
    """BASELINE.json configs[2] at the BENCHMARKED batch: 8 frames, 7 views, 64 planes, 120x160, 202-channel MLP; all 8
    frames against the oracle on a plane subset (the oracle's cost is per (pixel, plane)), masks of all frames exactly."""
    B = 8
    case = dict(B=B, K=7, C=16, D=64, h=120, w=160, seed=58)
    inp = synthetic.cost_volume_inputs(B, 7, 16, 120, 160, seed=case["seed"])
    mgr = _manager(case)
    mgr.volume_memory_format = torch.channels_last      # what DepthModel uses
    vol, lowest, planes, mask = _run(mgr, inp)
    planes_np = planes[:, :, 0, 0].cpu().numpy()
    sub = [0, 31, 63]
    cv_o, _, mask_o = _oracle(mgr, inp, planes_np[:, sub])
    for b in range(B):
        assert_close(vol[b:b + 1, sub], cv_o[b:b + 1], tol=2e-5, what=f"cfg3 B=8 frame {b} vs oracle")
    assert mismatch_fraction(mask, mask_o) == 0.0
    assert_lowest_cost(lowest, vol, planes_np, lowest.cpu().numpy(), "cfg3 B=8 self-consistency")


def test_cfg5_full_size_stress_config():
    """BASELINE.json configs[4] at FULL size: 15 source views (410-input MLP, W1 streamed from L2), 96 planes, 960x720
    images -> 180x240 matching maps, batch 4; metadata-MLP sweep against the oracle on a plane subset (all pixels, all 4
    frames), dot-product sweep against the oracle on all 96 planes of one frame."""
    B, K, C, D, h, w = 4, 15, 16, 96, 180, 240
    inp = synthetic.cost_volume_inputs(B, K, C, h, w, seed=91)
    mgr = _manager(dict(h=h, w=w, D=D, C=C, K=K, seed=17))
    assert mgr.mlp.net[0].in_features == 410
    vol, lowest, planes, mask = _run(mgr, inp)
    planes_np = planes[:, :, 0, 0].cpu().numpy()
    sub = [0, 47, 95]
    cv_o, _, mask_o = _oracle(mgr, inp, planes_np[:, sub])
    assert_close(vol[:, sub], cv_o, tol=2e-5, what="cfg5 full size vs oracle (planes 0, 47, 95)")
    assert mismatch_fraction(mask, mask_o) == 0.0
    assert_lowest_cost(lowest, vol, planes_np, lowest.cpu().numpy(), "cfg5 self-consistency")
    del vol
    from simplerecon_amd.cost_volume import CostVolumeManager
    one = {k: (v[:1].contiguous() if v.dim() > 0 and v.shape[0] == B else v) for k, v in inp.items()}
    dmgr = CostVolumeManager(h, w, num_depth_bins=D).to(DEV)
    with torch.inference_mode():
        dv, dlow, _, _ = dmgr(**{k: v.to(DEV) for k, v in one.items()})
    n = {k: v.numpy() for k, v in one.items()}
    dv_o, dlow_o, _ = oracle.dot_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["cur_invK"],
                                        planes_np[:1])
    assert_close(dv, dv_o, tol=2e-6, what="cfg5 dot sweep vs oracle")
    assert_lowest_cost(dlow, dv, planes_np[:1], dlow_o, "cfg5 dot sweep")
