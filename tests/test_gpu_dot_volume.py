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


# ------------------------------------------------------------------------------------------------------
# LDS-staged sweep (csrc/sr_dot_volume_lds.hip) against the L1-gather kernels (SR_DOT_LDS=0) and the oracle.
# The switches are read per call by the library, so they can be flipped inside this process.

class _env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _sweep_with_mask(inp, h, w, D, planes_bdhw=None):
    """sr_dot_volume_fwd through the C ABI with the mask output on (the manager never asks for it)."""
    from simplerecon_amd import _lib
    lib = _lib.lib()
    d = {k: v.to(DEV) for k, v in inp.items()}
    B, K, C = d["src_feats"].shape[:3]
    mgr = CostVolumeManager(h, w, num_depth_bins=D).to(DEV)
    planes = planes_bdhw.to(DEV) if planes_bdhw is not None else mgr.generate_depth_planes(B, d["min_depth"], d["max_depth"])
    vol = torch.full((B, D, h, w), float("nan"), device=DEV)
    lowest = torch.full((B, h, w), float("nan"), device=DEV)
    mask = torch.full((B, h, w), 7, dtype=torch.uint8, device=DEV)
    ws = torch.empty(lib.sr_volume_workspace_bytes(B, K, C, h, w), dtype=torch.uint8, device=DEV)
    rc = lib.sr_dot_volume_fwd(_lib.ptr(d["cur_feats"]), _lib.ptr(d["src_feats"]), _lib.ptr(d["src_Ks"]),
                               _lib.ptr(d["src_extrinsics"]), _lib.ptr(d["cur_invK"]), _lib.ptr(planes), *planes.stride(),
                               B, K, C, h, w, D, _lib.ptr(vol), D * h * w, h * w, 1, _lib.ptr(lowest), _lib.ptr(mask),
                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr(torch.device(DEV)))
    _lib.check(rc, "sr_dot_volume_fwd")
    torch.cuda.synchronize()
    return vol, lowest, mask, planes


def _zoomed(inp, zoom):
    """Source cameras with `zoom` x the focal length: zoom > 1 stretches a tile's footprint over zoom^2 as many texels."""
    out = dict(inp)
    Ks = inp["src_Ks"].clone()
    Ks[:, :, 0, 0] *= zoom
    Ks[:, :, 1, 1] *= zoom
    out["src_Ks"] = Ks
    return out


@pytest.mark.parametrize("shape", [dict(B=2, K=7, D=64, h=120, w=160), dict(B=1, K=3, D=13, h=37, w=53),
                                   dict(B=3, K=2, D=5, h=9, w=70), dict(B=1, K=1, D=1, h=8, w=32)])
@pytest.mark.parametrize("variant", [dict(SR_DOT_LDS_G=4), dict(SR_DOT_LDS_G=2), dict(SR_DOT_LDS_G=8, SR_DOT_LDS_CAP=770),
                                     dict(SR_DOT_LDS_G=4, SR_DOT_LDS_CULL=0)])
def test_lds_sweep_equals_gather_sweep(shape, variant):
    B, K, D, h, w = (shape[k] for k in "BKDhw")
    inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=70 + D)
    with _env(SR_DOT_LDS=0):
        v0, l0, m0, planes = _sweep_with_mask(inp, h, w, D)
    with _env(SR_DOT_LDS=1, **variant):
        v1, l1, m1, _ = _sweep_with_mask(inp, h, w, D)
    assert torch.isfinite(v1).all()
    assert rel_err(v1, v0) < 2e-6
    assert torch.equal(m1, m0)          # same projection arithmetic -> identical masks
    assert_lowest_cost(l1, v1, planes[:, :, 0, 0].cpu().numpy(), l0.cpu().numpy(), "lds vs gather")
    # where a whole view misses the image both kernels must produce exact zeros
    assert torch.equal(v1 == 0, v0 == 0)


@pytest.mark.parametrize("zoom", [0.3, 1.4, 2.0, 8.0])
def test_lds_sweep_footprint_split_and_global_fallback(zoom):
    """zoom 2: the 8-plane / 4-plane boxes no longer fit the LDS buffer and are split; zoom 8: not even one plane fits
    (taps straight from global memory); zoom 0.3: footprints of a few texels (many lanes share a tap)."""
    B, K, D, h, w = 1, 3, 16, 64, 96
    inp = _zoomed(synthetic.cost_volume_inputs(B, K, 16, h, w, seed=5), zoom)
    with _env(SR_DOT_LDS=1):
        v1, l1, m1, planes = _sweep_with_mask(inp, h, w, D)
    planes_np = planes[:, :, 0, 0].cpu().numpy()
    n = {k: v.numpy() for k, v in inp.items()}
    cv_o, low_o, mask_o = oracle.dot_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["cur_invK"],
                                            planes_np, want_mask=True)
    assert_close(v1, cv_o, tol=2e-6, what=f"zoom {zoom} vs oracle")
    assert_lowest_cost(l1, v1, planes_np, low_o, f"zoom {zoom}")
    assert np.array_equal(m1.cpu().numpy().astype(bool), mask_o.astype(bool))


def test_lds_sweep_per_pixel_planes_and_channels_last():
    """Caller-supplied per-pixel depth planes (cost_volume.py:247, 297-299) switch the hull culling off; channels-last
    volume strides."""
    B, K, D, h, w = 2, 4, 12, 40, 72
    inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=11)
    g = torch.Generator().manual_seed(3)
    planes = (0.3 + 4.0 * torch.rand((B, D, h, w), generator=g)).sort(dim=1).values.contiguous()
    with _env(SR_DOT_LDS=0):
        v0, l0, m0, _ = _sweep_with_mask(inp, h, w, D, planes)
    with _env(SR_DOT_LDS=1):
        v1, l1, m1, _ = _sweep_with_mask(inp, h, w, D, planes)
    assert rel_err(v1, v0) < 2e-6 and torch.equal(m1, m0)
    assert_lowest_cost(l1, v1, planes.numpy(), l0.cpu().numpy(), "per-pixel planes")
    n = {k: v.numpy() for k, v in inp.items()}
    cv_o, _, _ = oracle.dot_volume(n["cur_feats"], n["src_feats"], n["src_Ks"], n["src_extrinsics"], n["cur_invK"],
                                   planes.numpy())
    assert_close(v1, cv_o, tol=2e-6, what="per-pixel planes vs oracle")


def test_lowest_cost_treats_nan_like_torch_argmax():
    """A NaN cost is the maximum for torch.argmax (first NaN wins): reference cost_volume.py:374-378."""
    B, K, D, h, w = 1, 2, 16, 16, 64
    inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=2)
    inp["cur_feats"][0, 3, 5, 7] = float("nan")
    for lds in (0, 1):
        with _env(SR_DOT_LDS=lds):
            vol, lowest, _, planes = _sweep_with_mask(inp, h, w, D)
        idx = torch.argmax(vol, dim=1)
        want = torch.gather(planes, 1, idx[:, None])[:, 0]
        assert torch.isnan(vol[0, :, 5, 7]).any()
        assert torch.equal(lowest, want), f"SR_DOT_LDS={lds}"


def test_packed_reciprocal_is_the_ieee_division():
    from simplerecon_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    n = 1 << 20
    mant = rng.uniform(1.0, 2.0, n).astype(np.float32)
    expo = rng.integers(-60, 60, n)
    x = (np.ldexp(mant, expo) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    # neighbours of powers of two and of the range ends (hard cases for a Newton reciprocal)
    edge = np.array([1.0, 2.0, 0.5, 3.0, 2.0 ** -60, 2.0 ** 60, 1.9999999, 1.0000001, 0.99999994], dtype=np.float32)
    x[:edge.size] = edge
    x[edge.size:2 * edge.size] = -edge
    xs = torch.from_numpy(x).to(DEV)
    a, b = torch.empty_like(xs), torch.empty_like(xs)
    _lib.check(lib.sr_selftest_rcp(_lib.ptr(xs), _lib.ptr(a), _lib.ptr(b), n, _lib.stream_ptr(torch.device(DEV))), "rcp")
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert torch.equal(b.cpu(), 1.0 / torch.from_numpy(x))   # and the device division is the correctly rounded one
