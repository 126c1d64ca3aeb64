"""GPU parity of the fused dot-product sweep (sr_dot_volume_fwd via CostVolumeManager)
against the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle
from parity import assert_close, assert_lowest_cost, rel_err
from simplerecon_amd import synthetic
from simplerecon_amd.cost_volume import CostVolumeManager

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(case, inp, memory_format=torch.contiguous_format):
    mgr = CostVolumeManager(case["h"], case["w"], num_depth_bins=case["D"]).to(DEV)
    mgr.volume_memory_format = memory_format
    dinp = {k: v.to(DEV) for k, v in inp.items()}
    with torch.inference_mode():
        vol, lowest, planes, mask = mgr(return_mask=True, **dinp)
    torch.cuda.synchronize()
    assert mask is None  # reference cost_volume.py:286, 335
    return vol, lowest, planes


def _oracle(inp, planes_np):
    n = {k: v.numpy() for k, v in inp.items()}
    return oracle.dot_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["cur_invK"],
                             planes_np)


@pytest.mark.parametrize("name", [n for n, c in gc.VOLUME_CASES.items() if c["model"] == "dot"])
def test_dot_volume_matches_oracle_and_golden(name):
    case = gc.VOLUME_CASES[name]
    inp = gc.volume_inputs(case)
    gold = gc.load_golden("volume", name)
    vol, lowest, planes = _run(case, inp)
    planes_np = planes.cpu().numpy() if "depth_planes_bdhw" in inp else planes[:, :, 0, 0].cpu().numpy()
    if "depth_planes_bdhw" not in inp:
        assert_close(planes_np, gold["planes_bd"], tol=1e-6, what="depth planes")
    cv_o, low_o, _ = _oracle(inp, planes_np)
    # geometry is bit-identical to the oracle (contraction off), only the 64-term sums reassociate
    assert_close(vol, cv_o, tol=2e-6, what=f"{name} vs oracle")
    assert_close(vol, gold["cost_volume"], what=f"{name} vs reference golden")
    assert_lowest_cost(lowest, vol, planes_np, gold["lowest_cost"], name)
    assert_lowest_cost(lowest, vol, planes_np, low_o, name)


def test_channels_last_volume_is_the_same_volume():
    case = gc.VOLUME_CASES["dot_small"]
    inp = gc.volume_inputs(case)
    a, la, _ = _run(case, inp)
    b, lb, _ = _run(case, inp, torch.channels_last)
    assert b.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(a, b.contiguous()) and torch.equal(la, lb)


@pytest.mark.parametrize("B", [1, 3])
def test_cfg2_full_size(B):
    """BASELINE.json configs[1]: 7 source views, 64 planes, 640x480 -> 120x160 matching res."""
    case = dict(B=B, K=7, C=16, D=64, h=120, w=160, seed=40 + B)
    inp = synthetic.cost_volume_inputs(B, 7, 16, 120, 160, seed=case["seed"])
    vol, lowest, planes = _run(case, inp)
    planes_np = planes[:, :, 0, 0].cpu().numpy()
    cv_o, low_o, _ = _oracle(inp, planes_np)
    assert_close(vol, cv_o, tol=2e-6, what="cfg2 vs oracle")
    assert_lowest_cost(lowest, vol, planes_np, low_o, "cfg2")
    # size-independent properties ------------------------------------------------------
    # (1) linearity in the reference features
    inp2 = dict(inp)
    inp2["cur_feats"] = inp["cur_feats"] * 2.0
    vol2, _, _ = _run(case, inp2)
    assert torch.equal(vol2, vol * 2.0)
    # (2) the over-views reduction is a plain sum: permuting the source views changes only rounding
    perm = torch.tensor([3, 0, 6, 1, 5, 2, 4])
    inp3 = dict(inp)
    for k in ("src_feats", "src_extrinsics", "src_poses", "src_Ks"):
        inp3[k] = inp[k][:, perm].contiguous()
    vol3, _, _ = _run(case, inp3)
    assert rel_err(vol3, vol) < 2e-6
    # (3) frames in a batch are independent
    if B > 1:
        one = {k: (v[1:2].contiguous() if v.shape[0] == B else v) for k, v in inp.items()}
        v1, l1, _ = _run(dict(case, B=1), one)
        assert torch.equal(v1[0], vol[1]) and torch.equal(l1[0], lowest[1])


def test_ragged_and_empty():
    # width not a multiple of the 64-pixel wave tile, single plane, single view; empty batch
    case = dict(B=1, K=1, C=16, D=1, h=7, w=13, seed=9)
    inp = synthetic.cost_volume_inputs(1, 1, 16, 7, 13, seed=9)
    vol, lowest, planes = _run(case, inp)
    cv_o, low_o, _ = _oracle(inp, planes[:, :, 0, 0].cpu().numpy())
    assert_close(vol, cv_o, tol=2e-6, what="ragged")
    assert np.array_equal(lowest.cpu().numpy(), low_o)
    empty = {k: (v[:0] if v.dim() > 0 and v.shape[0] == 1 and k not in ("min_depth", "max_depth") else v)
             for k, v in inp.items()}
    vol, lowest, _ = _run(case, empty)
    assert vol.shape == (0, 1, 7, 13) and lowest.shape == (0, 7, 13)
