"""Pins the CPU oracle (oracle/, plain C) to the golden vectors produced by the
upstream reference's own modules (tests/golden/, generator make_golden.py).  CPU only."""
import numpy as np
import pytest

import golden_cases as gc
import oracle
from parity import assert_close, assert_lowest_cost, mismatch_fraction
from simplerecon_amd import synthetic


def _np(d):
    return {k: v.numpy() for k, v in d.items()}


def _planes(case, inp, gold):
    if "depth_planes_bdhw" in inp:
        return inp["depth_planes_bdhw"]
    return gold["planes_bd"]


def _mlp_weights(case):
    import torch
    from simplerecon_amd.networks import MLP
    cin = case["C"] * (case["K"] + 1) + 10 * case["K"] + 4
    mlp = synthetic.seeded_fill_(MLP([cin, 128, 128, 1], disable_final_activation=True), seed=case["seed"])
    sd = {k: v.numpy() for k, v in mlp.state_dict().items()}
    return dict(W1=sd["net.0.weight"], b1=sd["net.0.bias"], W2=sd["net.2.weight"], b2=sd["net.2.bias"],
                W3=sd["net.4.weight"], b3=sd["net.4.bias"])


@pytest.mark.parametrize("name", [n for n, c in gc.VOLUME_CASES.items() if c["model"] == "dot"])
def test_dot_volume_oracle_matches_reference(name):
    case = gc.VOLUME_CASES[name]
    inp = _np(gc.volume_inputs(case))
    gold = gc.load_golden("volume", name)
    planes = _planes(case, inp, gold)
    for prec in ("f32", "f64"):
        cv, low, mask = oracle.dot_volume(inp["cur_feats"], inp["src_feats"], inp["src_Ks"], inp["src_extrinsics"],
                                          inp["cur_invK"], planes, want_mask=True, precision=prec)
        assert_close(cv, gold["cost_volume"], what=f"{name}/{prec} cost_volume")
        assert_lowest_cost(low, cv, np.asarray(planes, dtype=cv.dtype), gold["lowest_cost"], name)


@pytest.mark.parametrize("name", [n for n, c in gc.VOLUME_CASES.items() if c["model"] == "hero"])
def test_mlp_volume_oracle_matches_reference(name):
    case = gc.VOLUME_CASES[name]
    inp = _np(gc.volume_inputs(case))
    gold = gc.load_golden("volume", name)
    planes = _planes(case, inp, gold)
    mlp = _mlp_weights(case)
    for prec in ("f32", "f64"):
        cv, low, mask = oracle.mlp_volume(inp["cur_feats"], inp["src_feats"], inp["src_Ks"], inp["src_extrinsics"],
                                          inp["src_poses"], inp["cur_invK"], planes, mlp, want_mask=True,
                                          precision=prec)
        assert_close(cv, gold["cost_volume"], what=f"{name}/{prec} feature volume")
        assert_close(cv, gold["cost_volume_fast"], what=f"{name}/{prec} vs FastFeatureVolumeManager")
        assert_lowest_cost(low, cv, np.asarray(planes, dtype=cv.dtype), gold["lowest_cost"], name)
        assert mismatch_fraction(mask, gold["overall_mask"]) == 0.0, f"{name}: overall_mask differs"


@pytest.mark.parametrize("name", ["hero_small", "hero_k7", "hero_edge"])
def test_mlp_input_channel_order(name):
    """The 202-vector layout (reference cost_volume.py:709-723), checked channel by channel
    against the tensor the reference actually fed to its MLP at the last depth plane."""
    case = gc.VOLUME_CASES[name]
    inp = _np(gc.volume_inputs(case))
    gold = gc.load_golden("volume", name)
    ref = gold["mlp_input_last_plane"]  # [B,h,w,Cin]
    d = float(gold["planes_bd"][0, -1])
    rng = np.random.default_rng(0)
    for _ in range(24):
        b, y, x = int(rng.integers(case["B"])), int(rng.integers(case["h"])), int(rng.integers(case["w"]))
        d = float(gold["planes_bd"][b, -1])
        vec = oracle.mlp_input(inp["cur_feats"], inp["src_feats"], inp["src_Ks"], inp["src_extrinsics"],
                               inp["src_poses"], inp["cur_invK"], d, b, y, x)
        want = ref[b, y, x]
        scale = np.maximum(np.abs(want), 1.0)
        assert (np.abs(vec - want) / scale).max() < 2e-5, (name, b, y, x, np.abs(vec - want).argmax())


@pytest.mark.parametrize("name", list(gc.BLOCK_CASES))
def test_basic_block_oracle(name):
    from simplerecon_amd.layers import BasicBlock
    case = gc.BLOCK_CASES[name]
    blk = synthetic.seeded_fill_(BasicBlock(case["cin"], case["cout"], stride=case["stride"]), seed=case["seed"])
    sd = {k: v.numpy() for k, v in blk.state_dict().items()}
    x = gc.block_input(case).numpy()
    gold = gc.load_golden("block", name)["out"]
    for prec in ("f32", "f64"):
        y = oracle.basic_block(x, sd, "", stride=case["stride"], precision=prec)
        assert_close(y, gold, what=f"BasicBlock {name}/{prec}")


def test_upsample_oracle():
    gold = np.load(gc.GOLDEN_DIR + "/upsample.npz")["out"]
    assert_close(oracle.upsample2x(gc.upsample_input().numpy()), gold, tol=1e-6, what="upsample2x")


@pytest.mark.parametrize("name", list(gc.NET_CASES))
def test_encoder_decoder_oracle(name):
    from simplerecon_amd.networks import CVEncoder, DepthDecoderPP
    case = gc.NET_CASES[name]
    enc = synthetic.seeded_fill_(CVEncoder(case["D"], case["enc_ch"][1:], case["cv_outs"]), seed=case["seed"])
    dec = synthetic.seeded_fill_(DepthDecoderPP(case["enc_ch"][:1] + case["cv_outs"]), seed=case["seed"] + 1)
    esd = {k: v.numpy() for k, v in enc.state_dict().items()}
    dsd = {k: v.numpy() for k, v in dec.state_dict().items()}
    vol, feats = gc.net_inputs(case)
    gold = gc.load_golden("net", name)
    cvf = oracle.cv_encoder(vol.numpy(), [f.numpy() for f in feats[1:]], esd)
    for i, t in enumerate(cvf):
        assert_close(t, gold[f"cv_feat_{i}"], what=f"{name} CVEncoder level {i}")
    outs = oracle.depth_decoder_pp([feats[0].numpy()] + cvf, dsd)
    for k, v in outs.items():
        assert_close(v, gold[k], what=f"{name} decoder {k}")


@pytest.mark.parametrize("name", list(gc.MATCHING_CASES))
def test_matching_encoder_oracle(name):
    """oracle.resnet_matching_encoder vs the reference's ResnetMatchingEncoder (networks.py:149-205) run on
    refshim's torch.nn restatement of the antialiased ResNet-18 stem."""
    from simplerecon_amd.networks import ResnetMatchingEncoder
    case = gc.MATCHING_CASES[name]
    enc = synthetic.seeded_fill_(ResnetMatchingEncoder(18, 16), seed=case["seed"])
    sd = {k: v.numpy() for k, v in enc.state_dict().items()}
    gold = gc.load_golden("matching", name)
    taps = {}
    out = oracle.resnet_matching_encoder(gc.matching_input(case).numpy(), sd, taps=taps)
    for k in ("stem", "pool", "layer1"):
        assert_close(taps[k], gold[k], what=f"{name} {k}")
    assert_close(out, gold["out"], what=f"{name} matching features")
    out64 = oracle.resnet_matching_encoder(gc.matching_input(case).numpy(), sd, precision="f64")
    assert_close(out64, gold["out"], what=f"{name} matching features (f64 arbitration)")


@pytest.mark.parametrize("name", list(gc.TSDF_CASES))
def test_tsdf_oracle_bit_exact(name):
    """oracle.tsdf_from_bounds / tsdf_integrate vs the reference's TSDF + TSDFFuser (tools/tsdf.py) run on CPU:
    voxel coordinates, TSDF values and weights are fp16 and must agree bit for bit."""
    case = gc.TSDF_CASES[name]
    gold = gc.load_golden("tsdf", name)
    origin, dims, coords, values, weights = oracle.tsdf_from_bounds(case["bounds"], case["voxel_size"])
    assert dims == gold["values"].shape
    assert np.array_equal(coords.view(np.uint16), gold["voxel_coords"].view(np.uint16))
    depth, K, T, mask = (t.numpy() for t in gc.tsdf_inputs(case))
    n1 = (case["frames"] + 1) // 2
    kw = dict(max_depth=case["max_depth"], voxel_size=case["voxel_size"])
    oracle.tsdf_integrate(values, weights, coords, depth[:n1], T[:n1], K[:n1], **kw)
    assert np.array_equal(values.view(np.uint16), gold["values_mid"].view(np.uint16))
    assert np.array_equal(weights.view(np.uint16), gold["weights_mid"].view(np.uint16))
    oracle.tsdf_integrate(values, weights, coords, depth[n1:], T[n1:], K[n1:], depth_mask_b1hw=mask[n1:], **kw)
    assert np.array_equal(values.view(np.uint16), gold["values"].view(np.uint16))
    assert np.array_equal(weights.view(np.uint16), gold["weights"].view(np.uint16))
    assert int((weights > 0).sum()) > 1000  # the case actually fuses something
