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
