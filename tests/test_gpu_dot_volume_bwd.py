"""Backward of the dot-product plane sweep (csrc/sr_dot_volume_bwd.hip through CostVolumeManager's autograd seam)
against the reference's own autograd (tests/golden/grad_dot.npz) and, at the full cfg2 size, against the exact
bilinearity of the volume: <d_cur, v> = L(cur + v) - L(cur) and <d_src, v> = L(src + v) - L(src) for
L = sum(cost_volume * R), evaluated with the (oracle-checked) HIP forward."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle
from parity import assert_close
from simplerecon_amd import synthetic
from simplerecon_amd.cost_volume import CostVolumeManager, FeatureVolumeManager

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grads(mgr, inp, R, memory_format=torch.contiguous_format):
    mgr.volume_memory_format = memory_format
    cur = inp["cur_feats"].clone().requires_grad_()
    src = inp["src_feats"].clone().requires_grad_()
    args = dict(inp, cur_feats=cur, src_feats=src)
    vol, lowest, _, mask = mgr(**args)
    assert vol.requires_grad and not lowest.requires_grad and mask is None
    (vol * R).sum().backward()
    return vol.detach(), cur.grad, src.grad


def test_gradients_match_reference_autograd():
    case = gc.GRAD_CASES["dot"]
    gold = gc.load_golden("grad", "dot")
    inp = {k: v.to(DEV) for k, v in gc.volume_inputs(case).items()}
    mgr = CostVolumeManager(case["h"], case["w"], num_depth_bins=case["D"]).to(DEV)
    R = torch.from_numpy(gc.grad_cotangent(case)).to(DEV)
    for fmt in (torch.contiguous_format, torch.channels_last):
        vol, d_cur, d_src = _grads(mgr, inp, R, fmt)
        assert_close(vol, gold["cost_volume"], what="forward under autograd")
        assert_close(d_cur, gold["d_cur_feats"], what=f"d cur_feats ({fmt})")
        assert_close(d_src, gold["d_src_feats"], what=f"d src_feats ({fmt})")


def test_only_requested_gradients_and_geometry_is_data():
    case = gc.GRAD_CASES["dot"]
    gold = gc.load_golden("grad", "dot")
    inp = {k: v.to(DEV) for k, v in gc.volume_inputs(case).items()}
    mgr = CostVolumeManager(case["h"], case["w"], num_depth_bins=case["D"]).to(DEV)
    R = torch.from_numpy(gc.grad_cotangent(case)).to(DEV)
    src = inp["src_feats"].clone().requires_grad_()
    vol = mgr(**dict(inp, src_feats=src))[0]
    (vol * R).sum().backward()
    assert_close(src.grad, gold["d_src_feats"], what="d src_feats alone")
    cur = inp["cur_feats"].clone().requires_grad_()
    vol = mgr(**dict(inp, cur_feats=cur))[0]
    vol.backward(R)
    assert_close(cur.grad, gold["d_cur_feats"], what="d cur_feats alone")
    with pytest.raises(NotImplementedError):
        mgr(**dict(inp, src_Ks=inp["src_Ks"].clone().requires_grad_()))
    # the metadata-MLP manager is differentiable out of the box too; opting out makes it refuse
    hero = FeatureVolumeManager(case["h"], case["w"], num_depth_bins=case["D"], num_source_views=case["K"]).to(DEV)
    assert hero(**dict(inp, cur_feats=cur))[0].requires_grad
    hero.differentiable = False
    with pytest.raises(NotImplementedError):
        hero(**dict(inp, cur_feats=cur))
    with torch.no_grad():
        assert not mgr(**dict(inp, cur_feats=cur))[0].requires_grad


@pytest.mark.parametrize("B,K,D,h,w", [(1, 7, 64, 120, 160), (2, 3, 5, 37, 29)])
def test_bilinearity_identity_full_size(B, K, D, h, w):
    inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=3, device=DEV)
    mgr = CostVolumeManager(h, w, num_depth_bins=D).to(DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    R = torch.randn((B, D, h, w), generator=g).to(DEV)
    _, d_cur, d_src = _grads(mgr, inp, R)
    assert torch.isfinite(d_cur).all() and torch.isfinite(d_src).all()

    def loss(**over):
        with torch.no_grad():
            return float((mgr(**dict(inp, **over))[0].double() * R.double()).sum())
    base = loss()
    for key, grad in (("cur_feats", d_cur), ("src_feats", d_src)):
        v = torch.randn(inp[key].shape, generator=g).to(DEV)
        want = float((grad.double() * v.double()).sum())
        got = loss(**{key: inp[key] + v}) - base
        scale = float(grad.double().norm() * v.double().norm()) / np.sqrt(v.numel())
        assert abs(got - want) <= 2e-4 * max(scale, abs(want)), (key, got, want, scale)


@pytest.mark.parametrize("case", [dict(B=2, K=3, D=5, h=37, w=29, seed=3), dict(B=1, K=4, D=6, h=20, w=28, seed=5, edge=True)])
def test_gradients_match_oracle(case):
    """Sizes / poses without reference goldens (ragged maps; a view behind the camera, identity pose, large rotation):
    against oracle.dot_volume_backward, itself pinned to the reference's autograd in tests/test_oracle_grad_golden.py."""
    case = dict(case, C=16, model="dot")
    inp = {k: v.to(DEV) for k, v in gc.volume_inputs(case).items()}
    mgr = CostVolumeManager(case["h"], case["w"], num_depth_bins=case["D"]).to(DEV)
    g = torch.Generator(device="cpu").manual_seed(9)
    R = torch.randn((case["B"], case["D"], case["h"], case["w"]), generator=g)
    _, d_cur, d_src = _grads(mgr, inp, R.to(DEV))
    n = {k: v.cpu().numpy() for k, v in inp.items()}
    planes = mgr.generate_depth_planes(case["B"], inp["min_depth"], inp["max_depth"])[:, :, 0, 0].cpu().numpy()
    o_cur, o_src = oracle.dot_volume_backward(R.numpy(), n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"],
                                              n["cur_invK"], planes)
    assert_close(d_cur, o_cur, what="d cur_feats vs oracle")
    assert_close(d_src, o_src, what="d src_feats vs oracle")
