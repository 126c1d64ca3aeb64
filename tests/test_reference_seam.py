"""The drop-in seam against the REAL reference classes (CPU, build container only: /root/reference is imported through
oracle/refshim.py and is absent on the GPU box, where these tests skip).

  * cost_volume.to_hip(reference manager) -- the attribute swap of INTEGRATION.md §1 / reference test.py:196-198 -- for all
    three reference managers: class, (C, K) recovered from the MLP width (reference cost_volume.py:420-435), MLP weights
    and linear_ramp_1d11 carried over, and the twin then runs through the C-ABI (stub library: argument marshalling).
  * reference checkpoints load with strict=True: CVEncoder, DepthDecoderPP, ResnetMatchingEncoder, the three managers
    (reference cost_volume.py:58-74, networks.py:20-127, 149-205), values included, and our state dicts load back into
    the reference's modules."""
import os

import pytest
import torch

import refshim
from simplerecon_amd import cost_volume as cv
from simplerecon_amd import networks as nets
from simplerecon_amd import synthetic
from test_host_logic_stub import stub  # noqa: F401  (fixture)

pytestmark = pytest.mark.skipif(not os.path.isdir(refshim.REFERENCE_ROOT), reason="reference checkout not present")

H, W, D = 12, 16, 8


@pytest.fixture(scope="module")
def ref():
    rcv, rnets, rlayers, rgeo, rgen = refshim.import_reference()
    return dict(cv=rcv, nets=rnets)


def _ref_manager(ref, kind, K):
    torch.manual_seed(3)
    if kind == "dot":
        return ref["cv"].CostVolumeManager(matching_height=H, matching_width=W, num_depth_bins=D)
    cls = ref["cv"].FeatureVolumeManager if kind == "mlp" else ref["cv"].FastFeatureVolumeManager
    # by value: the reference mutates the list it is given (cost_volume.py:402, 429)
    return cls(matching_height=H, matching_width=W, num_depth_bins=D, mlp_channels=[202, 128, 128, 1],
               matching_dim_size=16, num_source_views=K)


@pytest.mark.parametrize("kind,K", [("dot", 3), ("mlp", 7), ("mlp", 2), ("fast", 5)])
def test_to_hip_of_a_reference_manager(ref, stub, kind, K):  # noqa: F811
    m = _ref_manager(ref, kind, K)
    twin = cv.to_hip(m)
    assert (twin.matching_height, twin.matching_width, twin.num_depth_bins) == (H, W, D)
    assert torch.equal(twin.linear_ramp_1d11, m.linear_ramp_1d11)
    if kind == "dot":
        assert type(twin) is cv.CostVolumeManager and not hasattr(twin, "mlp")
    else:
        assert type(twin) is (cv.FastFeatureVolumeManager if kind == "fast" else cv.FeatureVolumeManager)
        # shared like the reference's own to_fast() (cost_volume.py:739-746): the same module, the same nn.Parameters
        assert twin.mlp is m.mlp
        assert [id(p) for p in twin.mlp.parameters()] == [id(p) for p in m.mlp.parameters()]
        assert (twin.matching_dim_size, twin.num_source_views) == (16, K)          # recovered from the MLP width
        assert twin.mlp_channels == [cv.mlp_input_channels(16, K), 128, 128, 1]
        sd_ref, sd = m.mlp.state_dict(), twin.mlp.state_dict()
        assert list(sd) == list(sd_ref)
        for k in sd:
            assert torch.equal(sd[k], sd_ref[k]), k
    # the twin's state dict has exactly the reference's keys and loads the reference's with strict=True
    assert sorted(twin.state_dict()) == sorted(m.state_dict())
    twin.load_state_dict(m.state_dict(), strict=True)
    # ... and it runs (stub library = every C-ABI call checked against the signature table)
    inp = synthetic.cost_volume_inputs(2, K, 16, H, W, seed=1)
    with torch.no_grad():
        vol, lowest, planes, mask = twin(return_mask=True, **inp)
    assert vol.shape == (2, D, H, W) and lowest.shape == (2, H, W) and planes.shape == (2, D, H, W)
    assert ("sr_dot_volume_fwd" if kind == "dot" else "sr_mlp_volume_fwd") in stub
    # depth planes = the reference's own generate_depth_planes
    want = m.generate_depth_planes(2, inp["min_depth"], inp["max_depth"])
    assert torch.allclose(planes, want, rtol=1e-6, atol=0)
    if kind != "dot":
        fast = twin.to_fast()   # the reference's second seam (cost_volume.py:739-746) on our side
        assert isinstance(fast, cv.FastFeatureVolumeManager) and fast.mlp is twin.mlp


def _assert_state_dicts_interchange(ours, theirs):
    a, b = ours.state_dict(), theirs.state_dict()
    assert list(a) == list(b), (set(a) ^ set(b))
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
    synthetic.seeded_fill_(theirs, seed=11)
    ours.load_state_dict(theirs.state_dict(), strict=True)
    for k, v in ours.state_dict().items():
        assert torch.equal(v, theirs.state_dict()[k]), k
    synthetic.seeded_fill_(ours, seed=12)
    theirs.load_state_dict(ours.state_dict(), strict=True)
    for k, v in theirs.state_dict().items():
        assert torch.equal(v, ours.state_dict()[k]), k


def test_reference_conv_stack_checkpoints_load_strictly(ref):
    enc_ch = [24, 48, 64, 160, 256]     # EfficientNetV2-S pyramid (reference depth_model.py:110-127)
    cv_out = [64, 128, 256, 384]
    _assert_state_dicts_interchange(nets.CVEncoder(num_ch_cv=D, num_ch_enc=enc_ch[1:], num_ch_outs=cv_out),
                                    ref["nets"].CVEncoder(num_ch_cv=D, num_ch_enc=enc_ch[1:], num_ch_outs=cv_out))
    dec_in = enc_ch[:1] + cv_out
    _assert_state_dicts_interchange(nets.DepthDecoderPP(dec_in), ref["nets"].DepthDecoderPP(dec_in))
    _assert_state_dicts_interchange(nets.MLP([202, 128, 128, 1], disable_final_activation=True),
                                    ref["nets"].MLP([202, 128, 128, 1], disable_final_activation=True))


def test_reference_matching_encoder_checkpoint_loads_strictly(ref):
    """Backbone = refshim's restatement of antialiased_cnns.resnet18 (package absent: SURVEY.md §8c); the tail and every
    state-dict key are the reference's (networks.py:163-201)."""
    _assert_state_dicts_interchange(nets.ResnetMatchingEncoder(18, 16).eval(),
                                    ref["nets"].ResnetMatchingEncoder(18, 16, pretrained=False).eval())


def test_pose_distance_on_host_tensors_matches_the_reference():
    """The reference's dataset workers call pose_distance on CPU poses (generic_mvs_dataset.py:643-659): host tensors take
    the host path and give the reference's values."""
    import numpy as np
    from simplerecon_amd import geometry
    rgeo = refshim.import_reference()[3]
    poses, _ = synthetic.poses(3, 5, seed=2)
    T = torch.from_numpy(poses.reshape(-1, 4, 4))
    ours, ref = geometry.pose_distance(T), rgeo.pose_distance(T)
    for a, b in zip(ours, ref):
        assert a.shape == b.shape and not a.is_cuda
        assert np.allclose(a.numpy(), b.numpy(), rtol=1e-6, atol=1e-7)
