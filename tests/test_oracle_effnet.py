"""CPU checks of the image-prior encoder restatement (oracle.efficientnetv2_s_features): cross-checked against a
second statement of the same public architecture written with ATen ops (tests/effnet_torch.py), plus the structural
anchors that are known without timm (parameter count, channel list, state-dict naming).  PARITY UNPINNED against the
reference's actual dependency (timm, unpinned in simplerecon_env.yml:22, absent here)."""
import numpy as np
import pytest
import torch

import effnet_torch
import oracle
from simplerecon_amd import synthetic
from simplerecon_amd.image_encoder import EfficientNetV2SFeatures


@pytest.fixture(scope="module")
def encoder():
    return synthetic.seeded_fill_(EfficientNetV2SFeatures(), seed=3, gain=1.0)


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 72, 88)])   # the second has odd maps at stride 8 and below
def test_oracle_matches_aten_restatement(encoder, shape):
    g = torch.Generator().manual_seed(shape[2])
    img = torch.randn(shape, generator=g)
    sd = encoder.state_dict()
    with torch.inference_mode():
        ref = effnet_torch.features(img, sd)
    out = oracle.efficientnetv2_s_features(img.numpy(), {k: v.numpy() for k, v in sd.items()})
    assert [r.shape[1] for r in ref] == [24, 48, 64, 160, 256]
    for i, (r, o) in enumerate(zip(ref, out)):
        r = r.numpy()
        assert r.shape == o.shape == (shape[0], r.shape[1], -(-shape[2] // (2 << i)), -(-shape[3] // (2 << i)))
        assert 0.1 < np.abs(r).max() < 1e3          # the seeded weights keep activations O(1) through 40 blocks
        assert np.abs(r - o).max() <= 1e-5 * np.abs(r).max(), i


def test_structure_anchors(encoder):
    sd = encoder.state_dict()
    n_params = sum(p.numel() for p in encoder.parameters())
    # timm reports 21 458 488 parameters for efficientnetv2_s with its 1000-class head; without conv_head
    # (256*1280), its BatchNorm (2*1280) and the classifier (1280*1000 + 1000) the feature stages hold:
    assert n_params == 21_458_488 - 256 * 1280 - 2 * 1280 - 1281 * 1000 == 19_847_248
    assert encoder.feature_info.channels() == encoder.num_ch_enc == [24, 48, 64, 160, 256]
    for key, shape in (("conv_stem.weight", (24, 3, 3, 3)), ("bn1.running_var", (24,)),
                       ("blocks.0.1.conv.weight", (24, 24, 3, 3)), ("blocks.1.0.conv_exp.weight", (96, 24, 3, 3)),
                       ("blocks.1.0.conv_pwl.weight", (48, 96, 1, 1)), ("blocks.2.3.bn2.weight", (64,)),
                       ("blocks.3.0.conv_pw.weight", (256, 64, 1, 1)), ("blocks.3.0.conv_dw.weight", (256, 1, 3, 3)),
                       ("blocks.3.0.se.conv_reduce.weight", (16, 256, 1, 1)),
                       ("blocks.3.5.se.conv_expand.bias", (512,)), ("blocks.4.0.conv_pwl.weight", (160, 768, 1, 1)),
                       ("blocks.4.8.se.conv_reduce.bias", (40,)), ("blocks.5.0.conv_dw.weight", (960, 1, 3, 3)),
                       ("blocks.5.14.bn3.running_mean", (256,)), ("blocks.5.14.se.conv_reduce.weight", (64, 1536, 1, 1))):
        assert tuple(sd[key].shape) == shape, key
    assert [len(s) for s in encoder.blocks] == [2, 4, 4, 6, 9, 15]
    assert all(m.eps == 1e-3 for m in encoder.modules() if isinstance(m, torch.nn.BatchNorm2d))


@pytest.mark.parametrize("case", [dict(cin=64, cout=64, stride=1, expand=4, hw=(12, 16), skip=True),
                                  dict(cin=48, cout=96, stride=2, expand=6, hw=(14, 18), skip=False),
                                  dict(cin=32, cout=32, stride=1, expand=6, hw=(9, 11), skip=True)])
def test_mbconv_block_matches_transformers_efficientnet_block(case):
    """The only INDEPENDENT statement of an MBConv + squeeze-excite block on this box: `transformers.models.efficientnet.
    modeling_efficientnet.EfficientNetBlock` (EfficientNet V1's block: expand 1x1 -> BN -> swish -> depthwise 3x3 with
    TF-"SAME" padding -> BN -> swish -> SE (reduce to 0.25 x block input, swish, expand, sigmoid) -> project 1x1 -> BN ->
    + input).  oracle.effnet_mbconv_block -- the function every stage-3..5 block of the image-prior encoder restatement
    goes through -- computes the same numbers from the same weights.  (Parity with timm's tf_efficientnetv2_s itself
    stays UNPINNED: timm is not installed here; this pins the block semantics, not the network definition.)"""
    M = pytest.importorskip("transformers.models.efficientnet.modeling_efficientnet")
    from transformers import EfficientNetConfig
    cfg = EfficientNetConfig(hidden_act="swish", batch_norm_eps=1e-3, squeeze_expansion_ratio=0.25)
    H, W = case["hw"]
    if case["stride"] == 2:
        assert H % 2 == 0 and W % 2 == 0   # transformers pads (0, 1, 0, 1) = TF-"SAME" only on even maps (adjust_padding)
    blk = M.EfficientNetBlock(cfg, in_dim=case["cin"], out_dim=case["cout"], stride=case["stride"],
                              expand_ratio=case["expand"], kernel_size=3, drop_rate=0.0, id_skip=not case["skip"],
                              adjust_padding=case["stride"] == 2).eval()
    g = torch.Generator().manual_seed(case["cin"] + case["cout"])
    with torch.no_grad():
        for name, p in blk.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1) + (1.0 if name.endswith("norm.weight") or
                                                                                         name.endswith("bn.weight") else 0.0))
        for name, b in blk.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
    x = torch.randn((2, case["cin"], H, W), generator=g)
    with torch.inference_mode():
        ref = blk(x).numpy()
    t = {k: v.detach().numpy() for k, v in blk.state_dict().items()}
    pre = "b."
    sd = {pre + "conv_pw.weight": t["expansion.expand_conv.weight"], pre + "conv_dw.weight": t["depthwise_conv.depthwise_conv.weight"],
          pre + "se.conv_reduce.weight": t["squeeze_excite.reduce.weight"], pre + "se.conv_reduce.bias": t["squeeze_excite.reduce.bias"],
          pre + "se.conv_expand.weight": t["squeeze_excite.expand.weight"], pre + "se.conv_expand.bias": t["squeeze_excite.expand.bias"],
          pre + "conv_pwl.weight": t["projection.project_conv.weight"]}
    for ours, theirs in (("bn1", "expansion.expand_bn"), ("bn2", "depthwise_conv.depthwise_norm"), ("bn3", "projection.project_bn")):
        for f in ("weight", "bias", "running_mean", "running_var"):
            sd[f"{pre}{ours}.{f}"] = t[f"{theirs}.{f}"]
    for precision, tol in (("f64", 1e-6), ("f32", 2e-5)):
        y = oracle.effnet_mbconv_block(x.numpy(), sd, pre, case["stride"], precision=precision)
        if case["skip"]:
            y = y + x.numpy()
        assert y.shape == ref.shape == (2, case["cout"], -(-H // case["stride"]), -(-W // case["stride"]))
        assert np.abs(y - ref).max() <= tol * np.abs(ref).max(), (precision, np.abs(y - ref).max() / np.abs(ref).max())
