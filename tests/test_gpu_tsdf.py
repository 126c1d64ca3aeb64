"""GPU parity of the TSDF fusion kernel (SURVEY.md §8f "next" #2; reference tools/tsdf.py:238-320): fp16 volumes
must agree BIT FOR BIT with the reference's TSDFFuser (golden, run on CPU) and with the numpy oracle."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import oracle
from simplerecon_amd import _lib
from simplerecon_amd.tsdf import TSDF, OurFuser, TSDFFuser

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bits(t):
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t
    return a.view(np.uint16)


@pytest.mark.parametrize("name", list(gc.TSDF_CASES))
def test_tsdf_vs_reference_golden(name):
    case = gc.TSDF_CASES[name]
    gold = gc.load_golden("tsdf", name)
    vol = TSDF.from_bounds(dict(case["bounds"]), voxel_size=case["voxel_size"], device=DEV)
    fuser = TSDFFuser(vol, max_depth=case["max_depth"])
    assert tuple(fuser.shape) == gold["values"].shape
    assert np.array_equal(_bits(vol.voxel_coords), _bits(gold["voxel_coords"]))
    depth, K, T, mask = (t.to(DEV) for t in gc.tsdf_inputs(case))
    n1 = (case["frames"] + 1) // 2
    fuser.integrate_depth(depth[:n1].half(), T[:n1].half(), K[:n1].half())
    assert np.array_equal(_bits(fuser.tsdf_values), _bits(gold["values_mid"]))
    assert np.array_equal(_bits(fuser.tsdf_weights), _bits(gold["weights_mid"]))
    fuser.integrate_depth(depth[n1:].half(), T[n1:].half(), K[n1:].half(), depth_mask_b1hw=mask[n1:])
    assert np.array_equal(_bits(fuser.tsdf_values), _bits(gold["values"]))
    assert np.array_equal(_bits(fuser.tsdf_weights), _bits(gold["weights"]))


def _random_scene(rng, frames=4):
    vs = float(rng.choice([0.04, 0.05, 0.0625, 0.03]))
    lo = rng.uniform(-1.5, -0.3, 3)
    hi = lo + rng.uniform(0.8, 2.2, 3)
    bounds = dict(xmin=lo[0], xmax=hi[0], ymin=lo[1], ymax=hi[1], zmin=lo[2], zmax=hi[2])
    H, W = int(rng.integers(20, 60)), int(rng.integers(24, 80))
    depth = (0.4 + 2.5 * rng.random((frames, 1, H, W))).astype(np.float32)
    K = np.tile(np.eye(4, dtype=np.float32), (frames, 1, 1))
    K[:, 0, 0], K[:, 1, 1] = rng.uniform(0.6, 1.4) * W, rng.uniform(0.6, 1.4) * W
    K[:, 0, 2], K[:, 1, 2] = W / 2 + rng.uniform(-3, 3), H / 2
    T = np.tile(np.eye(4, dtype=np.float32), (frames, 1, 1))
    for i in range(frames):
        q = rng.standard_normal(4)
        q[0] = abs(q[0]) + 1.5
        w_, x, y, z = q / np.linalg.norm(q)
        T[i, :3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_)],
                        [2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_)],
                        [2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)]]
        T[i, :3, 3] = rng.uniform(-0.5, 0.5, 3)
    return bounds, vs, depth, K, T, float(rng.choice([2.0, 3.0, 5.0])), float(rng.choice([0.5, 0.25, 0.3]))


@pytest.mark.parametrize("seed", range(6))
def test_tsdf_vs_oracle_random_scenes(seed):
    """Random volumes / cameras (cameras inside the volume, voxels on the camera plane, odd image sizes, scalars that
    are not representable in fp16): values and weights bit-identical to the oracle."""
    rng = np.random.default_rng(100 + seed)
    bounds, vs, depth, K, T, max_d, min_d = _random_scene(rng)
    vol = TSDF.from_bounds(dict(bounds), voxel_size=vs, device=DEV)
    fuser = TSDFFuser(vol, min_depth=min_d, max_depth=max_d)
    fuser.integrate_depth(*(torch.from_numpy(a).to(DEV).half() for a in (depth, T, K)))
    _, dims, coords, v, w = oracle.tsdf_from_bounds(bounds, vs)
    oracle.tsdf_integrate(v, w, coords, depth, T, K, min_depth=min_d, max_depth=max_d, voxel_size=vs)
    assert np.array_equal(_bits(vol.voxel_coords), coords.view(np.uint16))
    bad = (_bits(fuser.tsdf_values) != v.view(np.uint16)) | (_bits(fuser.tsdf_weights) != w.view(np.uint16))
    nan_both = np.isnan(fuser.tsdf_values.cpu().numpy()) & np.isnan(v)
    assert int((bad & ~nan_both).sum()) == 0, f"{int(bad.sum())} voxels differ of {int((w > 0).sum())} touched"


def test_tsdf_batch_equals_sequential_and_explicit_coords():
    """A batch of frames = the same frames one call at a time (the update is sequential per voxel), and a volume built
    from explicit voxel coordinates (TSDF(...) constructor) = the generated-coordinate fast path."""
    rng = np.random.default_rng(7)
    bounds, vs, depth, K, T, max_d, min_d = _random_scene(rng, frames=5)
    args = [torch.from_numpy(a).to(DEV).half() for a in (depth, T, K)]
    a = TSDFFuser(TSDF.from_bounds(dict(bounds), vs, device=DEV), min_d, max_d)
    a.integrate_depth(*args)
    b = TSDFFuser(TSDF.from_bounds(dict(bounds), vs, device=DEV), min_d, max_d)
    for i in range(5):
        b.integrate_depth(*(t[i:i + 1] for t in args))
    assert torch.equal(a.tsdf_values.view(torch.int16), b.tsdf_values.view(torch.int16))
    assert torch.equal(a.tsdf_weights.view(torch.int16), b.tsdf_weights.view(torch.int16))
    ref = TSDF.from_bounds(dict(bounds), vs, device=DEV)
    c = TSDFFuser(TSDF(ref.voxel_coords.clone(), ref.tsdf_values.clone(), ref.tsdf_weights.clone(), vs, ref.origin),
                  min_d, max_d)
    c.integrate_depth(*args)
    assert torch.equal(a.tsdf_values.view(torch.int16), c.tsdf_values.view(torch.int16))
    assert torch.equal(a.tsdf_weights.view(torch.int16), c.tsdf_weights.view(torch.int16))
    assert float(a.tsdf_weights.max()) <= 1.0 and int((a.tsdf_weights > 0).sum()) > 100


def test_tsdf_full_size_properties():
    """640x480 depth maps into a 6.4 m x 6.4 m x 3.2 m room at 4 cm (160x160x80 voxels): frames that cannot see the
    volume leave it untouched, weights stay in [0,1], untouched voxels keep (-1, 0)."""
    fuser = OurFuser(bounds=dict(xmin=-3.2, xmax=3.2, ymin=-3.2, ymax=3.2, zmin=0.0, zmax=3.2), max_fusion_depth=3.0,
                     device=DEV)
    f = fuser.tsdf_fuser_pred
    g = torch.Generator(device="cpu").manual_seed(3)
    depth = (1.0 + 1.5 * torch.rand((4, 1, 480, 640), generator=g)).to(DEV)
    K = torch.eye(4).repeat(4, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 577.87
    K[:, 0, 2], K[:, 1, 2] = 320.0, 240.0
    T_away = torch.eye(4).repeat(4, 1, 1)
    T_away[:, 2, 2], T_away[:, 0, 0] = -1.0, -1.0      # looking down -z: every voxel is behind the camera
    T_away[:, 2, 3] = -0.5
    fuser.fuse_frames(depth, K.to(DEV), T_away.to(DEV), None)
    assert bool((f.tsdf_values == -1).all()) and bool((f.tsdf_weights == 0).all())
    fuser.fuse_frames(depth, K.to(DEV), torch.eye(4).repeat(4, 1, 1).to(DEV), None)
    w, v = f.tsdf_weights.float(), f.tsdf_values.float()
    touched = w > 0
    assert 10000 < int(touched.sum()) < w.numel() // 2
    assert float(w.max()) <= 1.0 and float(w.min()) >= 0.0
    assert bool((v[~touched] == -1).all()) and float(v[touched].abs().max()) <= 1.0


def test_tsdf_refuses_wrong_inputs():
    vol = TSDF.from_bounds(dict(xmin=0, xmax=1, ymin=0, ymax=1, zmin=0, zmax=1), 0.05, device=DEV)
    f = TSDFFuser(vol)
    d, T, K = torch.ones(1, 1, 8, 8, device=DEV), torch.eye(4, device=DEV)[None], torch.eye(4, device=DEV)[None]
    with pytest.raises(TypeError):
        f.integrate_depth(d, T.half(), K.half())            # fp32 depth: the reference's fp16 matmul would fail too
    with pytest.raises(ValueError):
        f.integrate_depth(d.half()[:, 0], T.half(), K.half())
    with pytest.raises(_lib.HipLibraryError):
        TSDFFuser(vol, use_gpu=False)
    with pytest.raises(NotImplementedError):
        vol.to_mesh()
    vol.cpu()
    with pytest.raises(_lib.HipLibraryError):
        f.integrate_depth(d.half(), T.half(), K.half())
