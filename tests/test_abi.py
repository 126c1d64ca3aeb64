"""The C-ABI library loads and exports every symbol include/simplerecon_hip.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "simplerecon_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_exports_every_declared_symbol():
    from simplerecon_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -m simplerecon_amd.build` (or __graft_entry__.build())"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 4
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/simplerecon_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in simplerecon_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared, "binding lists symbols the header does not declare"


def test_binding_loads_and_reports_target():
    from simplerecon_amd import _lib
    lib = _lib.lib()
    assert lib.sr_abi_version() == _lib.ABI_VERSION
    assert lib.sr_target_arch() == b"gfx950"
    assert lib.sr_volume_workspace_bytes(1, 7, 16, 120, 160) >= 7 * 16 * 120 * 160 * 4


def test_missing_device_fails_loudly():
    """The product path never falls back to CPU: host tensors are rejected."""
    import torch
    from simplerecon_amd import synthetic
    from simplerecon_amd._lib import HipLibraryError
    from simplerecon_amd.cost_volume import CostVolumeManager
    mgr = CostVolumeManager(8, 12, num_depth_bins=4)
    inp = synthetic.cost_volume_inputs(1, 2, 16, 8, 12)
    with pytest.raises(HipLibraryError):
        mgr(**inp)


def test_argument_validation():
    import torch
    from simplerecon_amd.cost_volume import CostVolumeManager, FeatureVolumeManager, mlp_input_channels
    assert mlp_input_channels(16, 7) == 202 and mlp_input_channels(16, 2) == 72 and mlp_input_channels(16, 15) == 410
    m = FeatureVolumeManager(8, 12, num_depth_bins=4, num_source_views=2)
    assert m.mlp.net[0].in_features == 72
    # the reference's mutable-default quirk is NOT replicated
    m2 = FeatureVolumeManager(8, 12, num_depth_bins=4)
    assert m2.mlp.net[0].in_features == 202
    assert [k for k in CostVolumeManager(8, 12, 4).state_dict()] == [
        "linear_ramp_1d11", "backprojector.pix_coords_13N", "projector.eps"]


def test_matching_encoder_state_dict_names():
    """Checkpoint compatibility: the key set of the reference's ResnetMatchingEncoder(18, 16) state_dict
    (modules/networks.py:185-202 on the antialiased ResNet-18 backbone; verified against the reference class in the
    build container through oracle/refshim.py)."""
    from simplerecon_amd.networks import ResnetMatchingEncoder
    keys = set(ResnetMatchingEncoder(18, 16).state_dict())
    bn = ["weight", "bias", "running_mean", "running_var", "num_batches_tracked"]
    expect = {"net.0.weight", "net.3.1.filt", "net.5.weight", "net.5.bias", "net.8.weight", "net.8.bias"}
    expect |= {f"net.1.{k}" for k in bn}
    for blk in (0, 1):
        expect |= {f"net.4.{blk}.conv1.weight", f"net.4.{blk}.conv2.weight"}
        expect |= {f"net.4.{blk}.bn{i}.{k}" for i in (1, 2) for k in bn}
    assert keys == expect


def test_tf_same_padding_and_activation_codes():
    """Host logic of the image-prior encoder's conv wrappers: TF-"SAME" pads equal what timm's Conv2dSame computes
    (tests/effnet_torch._same), activation codes match include/simplerecon_hip.h."""
    import torch
    import effnet_torch
    from simplerecon_amd import ops
    for h, w in ((480, 640), (15, 20), (9, 11), (1, 7), (30, 1), (37, 51)):
        for k, s in ((3, 1), (3, 2), (1, 1)):
            pt, pl, pb, pr = ops.tf_same_pads(h, w, k, s)
            padded = effnet_torch._same(torch.zeros(1, 1, h, w), k, s)
            assert padded.shape[-2:] == (h + pt + pb, w + pl + pr)
            assert (h + pt + pb - k) // s + 1 == -(-h // s) and (w + pl + pr - k) // s + 1 == -(-w // s)
            assert pb - pt in (0, 1) and pr - pl in (0, 1)      # the odd pixel goes below / right
    hdr = open(os.path.join(ROOT, "include", "simplerecon_hip.h")).read()
    assert "#define SR_ACT_NONE (-1.0f)" in hdr and "#define SR_ACT_SILU (-2.0f)" in hdr
    assert ops._act_code(None, None) == -1.0 and ops._act_code(0.2, None) == 0.2 and ops._act_code(None, "silu") == -2.0
    with pytest.raises(ValueError):
        ops._act_code(0.2, "silu")
    with pytest.raises(ValueError):
        ops._act_code(None, "gelu")


def test_depth_model_default_construction_is_all_native():
    """DepthModel(opts) builds both encoders on the HIP modules, keeps the reference's sub-module names (checkpoint
    prefixes) and refuses host tensors (no CPU fallback)."""
    import torch
    from simplerecon_amd import depth_model as dm
    from simplerecon_amd._lib import HipLibraryError
    model = dm.DepthModel(dm.default_options(image_width=128, image_height=96, model_num_views=3,
                                             matching_num_depth_bins=8))
    assert type(model.encoder).__name__ == "EfficientNetV2SFeatures"
    assert type(model.matching_model).__name__ == "ResnetMatchingEncoder"
    prefixes = {k.split(".")[0] for k in model.state_dict()}
    assert prefixes == {"encoder", "cost_volume_net", "depth_decoder", "cost_volume", "matching_model"}
    with pytest.raises(HipLibraryError), torch.inference_mode():
        model.forward_tensors(torch.zeros(1, 3, 96, 128), torch.zeros(1, 2, 3, 96, 128), torch.eye(4).expand(1, 2, 4, 4),
                              torch.eye(4).expand(1, 2, 4, 4), torch.eye(4).expand(1, 2, 4, 4), torch.eye(4)[None])


def test_option_table_is_explicit_and_seeded_once_from_the_environment():
    """include/simplerecon_hip.h, SR_OPT_*: every run-time switch is an entry of one table -- named like the environment
    variable that seeds it at the first access, read / set through the C ABI afterwards (host-only code: runs without a GPU).
    Changing the environment later has no effect."""
    import ctypes as C
    import os
    from simplerecon_amd import _lib
    lib = _lib.lib()
    n = lib.sr_option_count()
    names = [lib.sr_option_name(i).decode() for i in range(n)]
    assert len(set(names)) == n and all(nm.startswith("SR_") for nm in names)
    assert {"SR_MLP_SPLIT", "SR_WINO_SPLIT", "SR_WINO_XCD", "SR_PW_NT", "SR_DOT_LDS"} <= set(names)
    assert lib.sr_option_id(b"SR_NO_SUCH_SWITCH") == -1 and lib.sr_option_name(n) is None
    for i, nm in enumerate(names):
        assert lib.sr_option_id(nm.encode()) == i
    d = C.c_int(0)
    assert lib.sr_option_default(lib.sr_option_id(b"SR_WINO_XCD"), C.byref(d)) == 0 and d.value == 1
    before = _lib.get_option("SR_PW_NT")
    os.environ["SR_PW_NT"] = "4"          # too late: the table was seeded at the first access
    try:
        assert _lib.get_option("SR_PW_NT") == before
        with _lib.option("SR_PW_NT", 2):
            assert _lib.get_option("SR_PW_NT") == 2
            with _lib.option("SR_PW_NT", 1):
                assert _lib.get_option("SR_PW_NT") == 1
            assert _lib.get_option("SR_PW_NT") == 2
        assert _lib.get_option("SR_PW_NT") == before
    finally:
        del os.environ["SR_PW_NT"]
    assert _lib.set_option("SR_MLP_SPLIT", "int8") in (0, 1, 2, -1) and _lib.get_option("SR_MLP_SPLIT") == -1
    _lib.set_option("SR_MLP_SPLIT", 0)
    assert lib.sr_option_set(n, 0, None) == 1 and lib.sr_option_get(-1, C.byref(d)) == 1   # SR_ERR_INVALID_ARGUMENT
