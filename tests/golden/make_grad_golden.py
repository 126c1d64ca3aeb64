"""Generates tests/golden/grad_*.npz: gradients of the reference's cost-volume managers, from the reference's own
autograd on CPU (build container only; needs /root/reference):

    python tests/golden/make_grad_golden.py

Groundwork for SURVEY.md §8f "next" #3 (backward pass of the fused cost volume): loss = sum(cost_volume * R) with a
seeded cotangent R; stored are dL/d cur_feats, dL/d src_feats and (hero model) dL/d of the six MLP tensors.  Inputs,
weights and R are regenerated from (seed, shape): see tests/golden_cases.py::GRAD_CASES / grad_cotangent.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import refshim  # noqa: E402
import golden_cases as gc  # noqa: E402
from simplerecon_amd import synthetic  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(8)
    cv, nets, layers = refshim.import_reference()[:3]
    # ---- CVEncoder (concat, stride-2 levels, chained blocks) --------------------------------------------
    case = gc.NET_CASES["narrow"]
    enc = nets.CVEncoder(num_ch_cv=case["D"], num_ch_enc=case["enc_ch"][1:], num_ch_outs=case["cv_outs"])
    synthetic.seeded_fill_(enc, seed=case["seed"])
    vol, feats = gc.net_inputs(case)
    vol.requires_grad_()
    feats = [f.requires_grad_() for f in feats]
    outs = enc(vol, feats[1:])
    cot = gc.cv_encoder_cotangents(case, [tuple(o.shape) for o in outs])
    sum((o * torch.from_numpy(c)).sum() for o, c in zip(outs, cot)).backward()
    save = {"d_x": vol.grad.numpy()}
    save.update({f"d_feat_{i + 1}": f.grad.numpy() for i, f in enumerate(feats[1:])})
    keep = ("convs.ds_conv_0.conv1.weight", "convs.ds_conv_1.downsample.0.weight", "convs.conv_0.0.downsample.0.weight",
            "convs.conv_1.1.conv2.bias", "convs.conv_3.1.conv2.bias", "convs.ds_conv_3.conv1.bias")
    prm = dict(enc.named_parameters())
    save.update({"d_" + k: prm[k].grad.numpy() for k in keep})
    np.savez_compressed(os.path.join(OUT, "grad_cv_encoder_narrow.npz"), **save)
    print("cv_encoder", {k: v.shape for k, v in save.items()})
    # ---- DepthDecoderPP (UNet++ graph, bilinear x2, heads) -------------------------------------------------
    # Run in float64: with ~2 M LeakyReLU pre-activations in this graph a handful fall within fp32 rounding of the
    # kink, and which side they land on depends on the summation order of the convolution -- an fp32 run differs
    # from ANY other implementation by ~2e-3 (relative L2) for that reason alone.  The float64 run pins the
    # structure exactly (the oracle matches it to 1e-15).
    dec = nets.DepthDecoderPP(case["enc_ch"][:1] + case["cv_outs"])
    synthetic.seeded_fill_(dec, seed=case["seed"] + 1)
    dec = dec.double()
    dec_in = [t.double().requires_grad_() for t in gc.decoder_inputs(case)]
    outs = dec(dec_in)
    cot = gc.decoder_cotangents(case, {k: tuple(v.shape) for k, v in outs.items()})
    sum((outs[k] * torch.from_numpy(c).double()).sum() for k, c in cot.items()).backward()
    save = {f"d_feat_{i}": t.grad.numpy() for i, t in enumerate(dec_in)}
    prm = dict(dec.named_parameters())
    save.update({"d_" + k: prm[k].grad.numpy() for k in gc.DECODER_GRAD_PARAMS})
    np.savez_compressed(os.path.join(OUT, "grad_decoder_narrow.npz"), **save)
    print("decoder", {k: v.shape for k, v in save.items()})
    # ---- the whole training chain behind the encoders, float64 (see the decoder note above) ---------------
    case = gc.CHAIN_CASE
    inp = synthetic.cost_volume_inputs(case["B"], case["K"], case["C"], case["h"], case["w"], seed=case["seed"], device="cpu")
    mgr = cv.FeatureVolumeManager(case["h"], case["w"], num_depth_bins=case["D"],
                                  mlp_channels=[case["C"] * (case["K"] + 1) + 10 * case["K"] + 4, 128, 128, 1],
                                  matching_dim_size=case["C"], num_source_views=case["K"])
    synthetic.seeded_fill_(mgr.mlp, seed=case["seed"])
    enc = synthetic.seeded_fill_(nets.CVEncoder(num_ch_cv=case["D"], num_ch_enc=case["enc_ch"][1:],
                                                num_ch_outs=case["cv_outs"]), seed=case["seed"] + 1)
    dec = synthetic.seeded_fill_(nets.DepthDecoderPP(case["enc_ch"][:1] + case["cv_outs"]), seed=case["seed"] + 2)
    mgr, enc, dec = mgr.double(), enc.double(), dec.double()
    inp = {k: (v.double() if v.is_floating_point() else v) for k, v in inp.items()}
    inp["cur_feats"].requires_grad_()
    inp["src_feats"].requires_grad_()
    pyr = [f.double().requires_grad_() for f in synthetic.image_prior_pyramid(case["B"], case["h"], case["w"],
                                                                             chans=case["enc_ch"], seed=case["seed"])]
    vol = mgr(**inp)[0]
    outs = dec(pyr[:1] + enc(vol, pyr[1:]))
    cot = gc.chain_cotangents(case, {k: tuple(v.shape) for k, v in outs.items()})
    sum((torch.exp(outs[k]) * torch.from_numpy(c).double()).sum() for k, c in cot.items()).backward()
    save = {"depth_s0": torch.exp(outs["log_depth_pred_s0_b1hw"]).detach().numpy(), "cost_volume": vol.detach().numpy(),
            "d_cur_feats": inp["cur_feats"].grad.numpy(), "d_src_feats": inp["src_feats"].grad.numpy()}
    save.update({f"d_pyr_{i}": t.grad.numpy() for i, t in enumerate(pyr)})
    save.update({"d_mlp." + k: v.grad.numpy() for k, v in mgr.mlp.state_dict(keep_vars=True).items()})
    prm = dict(enc.named_parameters())
    save.update({"d_enc." + k: prm[k].grad.numpy() for k in gc.CHAIN_ENC_PARAMS})
    prm = dict(dec.named_parameters())
    save.update({"d_dec." + k: prm[k].grad.numpy() for k in gc.DECODER_GRAD_PARAMS})
    np.savez_compressed(os.path.join(OUT, "grad_chain_narrow.npz"), **save)
    print("chain", {k: v.shape for k, v in save.items()})
    # ---- BasicBlock (conv stack) ---------------------------------------------------------------------
    for name in gc.GRAD_BLOCK_CASES:
        case = gc.BLOCK_CASES[name]
        blk = layers.BasicBlock(case["cin"], case["cout"], stride=case["stride"])
        synthetic.seeded_fill_(blk, seed=case["seed"])
        x = gc.block_input(case).requires_grad_()
        y = blk(x)
        R = torch.from_numpy(gc.block_cotangent(case, tuple(y.shape)))
        (y * R).sum().backward()
        out = {"out": y.detach().numpy(), "d_x": x.grad.numpy()}
        out.update({"d_" + k: p.grad.numpy() for k, p in blk.named_parameters()})
        np.savez_compressed(os.path.join(OUT, f"grad_block_{name}.npz"), **out)
        print("block", name, {k: v.shape for k, v in out.items()})
    for name, case in gc.GRAD_CASES.items():
        inp = gc.volume_inputs(case)
        h, w, D, K, C = case["h"], case["w"], case["D"], case["K"], case["C"]
        if case["model"] == "dot":
            mgr = cv.CostVolumeManager(h, w, num_depth_bins=D)
        else:
            mgr = cv.FeatureVolumeManager(h, w, num_depth_bins=D, mlp_channels=[C * (K + 1) + 10 * K + 4, 128, 128, 1],
                                          matching_dim_size=C, num_source_views=K)
            synthetic.seeded_fill_(mgr.mlp, seed=case["seed"])
        inp["cur_feats"].requires_grad_()
        inp["src_feats"].requires_grad_()
        vol = mgr(**inp)[0]
        R = torch.from_numpy(gc.grad_cotangent(case))
        (vol * R).sum().backward()
        out = dict(cost_volume=vol.detach().numpy(), d_cur_feats=inp["cur_feats"].grad.numpy(),
                   d_src_feats=inp["src_feats"].grad.numpy())
        if case["model"] == "hero":
            for k, prm in mgr.mlp.state_dict(keep_vars=True).items():
                out["d_mlp." + k] = prm.grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"grad_{name}.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
