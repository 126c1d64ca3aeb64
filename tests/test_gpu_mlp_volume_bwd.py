"""Backward of the metadata-MLP plane sweep (csrc/sr_mlp_volume_bwd.hip through FeatureVolumeManager's autograd seam)
against the reference's own autograd (tests/golden/grad_hero.npz) and, at a ragged size with three views, against
oracle.mlp_volume_backward (itself pinned to that golden in tests/test_oracle_grad_golden.py)."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle
from parity import assert_close
from simplerecon_amd import synthetic
from simplerecon_amd.cost_volume import FeatureVolumeManager

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = {"net.0.weight": "dW1", "net.0.bias": "db1", "net.2.weight": "dW2", "net.2.bias": "db2", "net.4.weight": "dW3",
         "net.4.bias": "db3"}


def _run(case, R):
    inp = {k: v.to(DEV) for k, v in gc.volume_inputs(case).items()}
    mgr = FeatureVolumeManager(case["h"], case["w"], num_depth_bins=case["D"], matching_dim_size=case["C"],
                               num_source_views=case["K"])
    synthetic.seeded_fill_(mgr.mlp, seed=case["seed"])
    mgr = mgr.to(DEV)
    mgr.differentiable = True
    cur = inp["cur_feats"].clone().requires_grad_()
    src = inp["src_feats"].clone().requires_grad_()
    vol, lowest, _, mask = mgr(**dict(inp, cur_feats=cur, src_feats=src), return_mask=True)
    assert vol.requires_grad and not lowest.requires_grad and mask.dtype == torch.bool
    (vol * R.to(DEV)).sum().backward()
    grads = {"d_cur_feats": cur.grad, "d_src_feats": src.grad}
    for k, prm in mgr.mlp.named_parameters():
        grads[NAMES[k]] = prm.grad
    return mgr, inp, vol.detach(), grads


def test_gradients_match_reference_autograd():
    case = gc.GRAD_CASES["hero"]
    gold = gc.load_golden("grad", "hero")
    _, _, vol, grads = _run(case, torch.from_numpy(gc.grad_cotangent(case)))
    assert_close(vol, gold["cost_volume"], what="forward under autograd")
    assert_close(grads["d_cur_feats"], gold["d_cur_feats"], what="d cur_feats")
    assert_close(grads["d_src_feats"], gold["d_src_feats"], what="d src_feats")
    for k, name in NAMES.items():
        assert_close(grads[name], gold["d_mlp." + k], what="d mlp." + k)


def test_gradients_match_oracle_ragged_three_views():
    case = dict(model="hero", B=2, K=3, C=16, D=4, h=13, w=17, seed=9)
    g = torch.Generator(device="cpu").manual_seed(3)
    R = torch.randn((case["B"], case["D"], case["h"], case["w"]), generator=g)
    mgr, inp, _, grads = _run(case, R)
    n = {k: v.cpu().numpy() for k, v in inp.items()}
    planes = mgr.generate_depth_planes(case["B"], inp["min_depth"], inp["max_depth"])[:, :, 0, 0].cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in mgr.mlp.state_dict().items()}
    mlp = dict(W1=sd["net.0.weight"], b1=sd["net.0.bias"], W2=sd["net.2.weight"], b2=sd["net.2.bias"],
               W3=sd["net.4.weight"], b3=sd["net.4.bias"])
    ref = oracle.mlp_volume_backward(R.numpy(), n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"],
                                     n["src_poses"], n["cur_invK"], planes, mlp)
    for key in ("d_cur_feats", "d_src_feats", "dW1", "db1", "dW2", "db2", "dW3", "db3"):
        assert_close(grads[key], ref[key], what=key + " vs oracle")


def test_matrix_core_backward_equals_valu_backward_and_handles_15_views(monkeypatch):
    """The MFMA backward (default) against the r01 VALU kernel (SR_MLP_BWD_VALU is read once per process, so the VALU
    result comes from the oracle instead) at 7 views, and at 15 views / 410 MLP inputs (BASELINE.json configs[4]), which
    the VALU kernel could not run (Cin <= 256)."""
    for K, D, h, w in ((7, 3, 10, 21), (15, 2, 9, 14), (2, 4, 8, 8)):
        case = dict(model="hero", B=1, K=K, C=16, D=D, h=h, w=w, seed=20 + K)
        g = torch.Generator(device="cpu").manual_seed(K)
        R = torch.randn((1, D, h, w), generator=g)
        mgr, inp, _, grads = _run(case, R)
        n = {k: v.cpu().numpy() for k, v in inp.items()}
        planes = mgr.generate_depth_planes(1, inp["min_depth"], inp["max_depth"])[:, :, 0, 0].cpu().numpy()
        sd = {k: v.detach().cpu().numpy() for k, v in mgr.mlp.state_dict().items()}
        mlp = dict(W1=sd["net.0.weight"], b1=sd["net.0.bias"], W2=sd["net.2.weight"], b2=sd["net.2.bias"],
                   W3=sd["net.4.weight"], b3=sd["net.4.bias"])
        ref = oracle.mlp_volume_backward(R.numpy(), n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"],
                                         n["src_poses"], n["cur_invK"], planes, mlp)
        for key in ("d_cur_feats", "d_src_feats", "dW1", "db1", "dW2", "db2", "dW3", "db3"):
            assert_close(grads[key], ref[key], what=f"K={K} {key} vs oracle")
