"""Training path of the two encoders (SURVEY.md §8f "next" #3; reference train.py:126-145 differentiates
modules/networks.py:149-205 and the timm EfficientNetV2-S pyramid with BatchNorm in training mode).

 * operator level: every differentiable operator of simplerecon_amd/train_ops.py (HIP forward + backward) against the
   same operator in ATen on CPU (float64 autograd);
 * ResnetMatchingEncoder: forward output, EVERY parameter gradient and the BatchNorm running-statistics update against
   the reference's own class and autograd (tests/golden/grad_matching_encoder_{train,eval}.npz, float64);
 * EfficientNetV2SFeatures: pyramid and parameter gradients against the ATen restatement of the same public definition
   (tests/effnet_torch.py; timm itself is absent -- parity unpinned), training and eval BatchNorm;
 * DepthModel.forward under autograd: every parameter of every stage receives a finite gradient; freeze_encoders opts out.

Deep-graph gradients are compared in relative L2 (ReLU / max-pool decisions sit on fp32 rounding for a handful of
elements); operator-level checks are element-wise (range-relative 1e-4 / 2e-5)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import effnet_torch
import golden_cases as gc
from parity import rel_err
from simplerecon_amd import depth_model as dm
from simplerecon_amd import image_encoder, networks, ops, synthetic
from simplerecon_amd import train_ops as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b, floor=0.0):
    """||a - b|| / max(||b||, floor * sqrt(n)).  `floor` = the magnitude below which a gradient counts as zero: a bias in
    front of an InstanceNorm / a training-mode BatchNorm has an exactly zero gradient (1e-17 in the float64 golden, 1e-9
    of fp32 cancellation noise here) and no meaningful relative error."""
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), floor * np.sqrt(a.size), 1e-30))


def _nhwc(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def _randn(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32))


# ------------------------------------------------------------------------------------------- operators ----------
@pytest.mark.parametrize("mode,act", [(m, a) for m in ("bn_train", "bn_eval") for a in (T.ACT_NONE, 0.0, T.ACT_SILU)] +
                         [("in_leaky", T.ACT_NONE), ("in_plain", T.ACT_NONE)])   # InstanceNorm carries its own LeakyReLU
def test_norm_act_forward_backward(mode, act):
    B, Cn, H, W = 3, 24, 9, 13
    x0, cot = _randn((B, Cn, H, W), 1) * 1.5 + 0.3, _randn((B, Cn, H, W), 2)
    bn = torch.nn.BatchNorm2d(Cn, eps=1e-3)
    synthetic.seeded_fill_(bn, seed=3)
    bn.train(mode == "bn_train")

    def torch_act(z):
        return z if act == T.ACT_NONE else (F.relu(z) if act == 0.0 else F.silu(z))
    # ATen reference in float64
    xr = x0.double().requires_grad_()
    bnr = torch.nn.BatchNorm2d(Cn, eps=1e-3).double()
    bnr.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    bnr.train(mode == "bn_train")
    if mode.startswith("bn"):
        yr = torch_act(bnr(xr))
    else:
        z = F.instance_norm(xr, eps=1e-5)
        yr = F.leaky_relu(z, 0.2) if mode == "in_leaky" else z
    (yr * cot.double()).sum().backward()
    # HIP
    bn = bn.to(DEV)
    x = _nhwc(x0).requires_grad_()
    if mode.startswith("bn"):
        y = T.batch_norm_act(x, bn, act=act)
    else:
        y = T.instance_norm_act(x, eps=1e-5, leaky=0.2 if mode == "in_leaky" else None)
    (y * cot.to(DEV)).sum().backward()
    assert rel_err(y, yr.detach()) < 2e-5
    assert rel_err(x.grad, xr.grad) < 1e-4
    if mode.startswith("bn"):
        assert rel_err(bn.weight.grad, bnr.weight.grad) < 1e-4 and rel_err(bn.bias.grad, bnr.bias.grad) < 1e-4
        # running statistics: updated in training mode (momentum 0.1, unbiased variance), untouched in eval mode
        assert rel_err(bn.running_mean, bnr.running_mean) < 1e-5 and rel_err(bn.running_var, bnr.running_var) < 1e-5
        assert int(bn.num_batches_tracked) == int(bnr.num_batches_tracked)


@pytest.mark.parametrize("shape", [(2, 8, 12, 16), (1, 64, 9, 11), (2, 4, 4, 4), (1, 12, 33, 18)])
def test_maxblurpool_backward(shape):
    B, Cn, H, W = shape
    x0, = (_randn(shape, 5),)
    xr = x0.double().requires_grad_()
    m = F.max_pool2d(xr, 2, stride=1)
    a = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64)
    filt = (a[:, None] * a[None, :] / 64.0)[None, None].repeat(Cn, 1, 1, 1)
    yr = F.conv2d(F.pad(m, (1, 2, 1, 2), mode="reflect"), filt, stride=2, groups=Cn)
    cot = _randn(tuple(yr.shape), 6)
    (yr * cot.double()).sum().backward()
    x = _nhwc(x0).requires_grad_()
    y = T.maxblurpool(x)
    (y * cot.to(DEV)).sum().backward()
    assert rel_err(y, yr.detach()) < 2e-6
    assert rel_err(x.grad, xr.grad) < 2e-6


@pytest.mark.parametrize("cfg", [(2, 16, 24, 10, 14, 3, 1, None), (2, 24, 48, 12, 16, 3, 2, "same"), (1, 32, 16, 9, 13, 3, 2, "same"),
                                 (2, 128, 16, 7, 9, 3, 1, "valid_rep"), (3, 64, 128, 6, 10, 1, 1, None), (1, 3, 24, 16, 20, 3, 2, "same")])
def test_conv_with_explicit_pads_forward_backward(cfg):
    """TF-"SAME" stride-2 convs (asymmetric zero padding) and the valid convolution behind a replicate pad."""
    B, ci, co, H, W, k, s, kind = cfg
    torch.manual_seed(ci + co)   # fixed module init: the checks must not depend on test order
    conv = torch.nn.Conv2d(ci, co, k, stride=s, padding=k // 2, bias=(kind != "same"))
    x0 = _randn((B, ci, H, W), 7)
    xr = x0.double().requires_grad_()
    cr = torch.nn.Conv2d(ci, co, k, stride=s, padding=0, bias=conv.bias is not None).double()
    cr.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    if kind == "same":
        pads = ops.tf_same_pads(H, W, k, s)
        yr = cr(F.pad(xr, (pads[1], pads[3], pads[0], pads[2])))
    elif kind == "valid_rep":
        pads = (0, 0, 0, 0)
        yr = cr(F.pad(xr, (1, 1, 1, 1), mode="replicate"))
    else:
        pads = None
        yr = cr(F.pad(xr, (k // 2,) * 4))
    cot = _randn(tuple(yr.shape), 8)
    (yr * cot.double()).sum().backward()
    conv = conv.to(DEV)
    x = (_nhwc(x0) if ci > 3 else x0.to(DEV)).requires_grad_()
    xin = T.replicate_pad(x, 1) if kind == "valid_rep" else x
    y = T.conv(xin, conv, pads=pads)
    (y * cot.to(DEV)).sum().backward()
    assert rel_err(y, yr.detach()) < 2e-5
    assert rel_err(x.grad, xr.grad) < 1e-4
    assert rel_err(conv.weight.grad, cr.weight.grad) < 1e-4
    if conv.bias is not None:
        assert rel_err(conv.bias.grad, cr.bias.grad) < 1e-4


@pytest.mark.parametrize("cfg", [(2, 64, 12, 16, 1), (2, 96, 12, 16, 2), (1, 512, 7, 9, 2), (3, 128, 5, 8, 1)])
def test_depthwise_and_squeeze_excite(cfg):
    B, Cn, H, W, s = cfg
    rd = max(4, Cn // 16)
    torch.manual_seed(Cn)
    dw = torch.nn.Conv2d(Cn, Cn, 3, stride=s, padding=1, groups=Cn, bias=False)
    r, e = torch.nn.Conv2d(Cn, rd, 1), torch.nn.Conv2d(rd, Cn, 1)
    x0 = _randn((B, Cn, H, W), 9)
    pads = ops.tf_same_pads(H, W, 3, s)
    xr = x0.double().requires_grad_()
    mods_r = [m_.__class__(*a, **kw).double() for m_, a, kw in
              ((dw, (Cn, Cn, 3), dict(stride=s, padding=0, groups=Cn, bias=False)), (r, (Cn, rd, 1), {}), (e, (rd, Cn, 1), {}))]
    for mr, m_ in zip(mods_r, (dw, r, e)):
        mr.load_state_dict({n: v.double() for n, v in m_.state_dict().items()})
    d = mods_r[0](F.pad(xr, (pads[1], pads[3], pads[0], pads[2])))
    g = torch.sigmoid(mods_r[2](F.silu(mods_r[1](d.mean((2, 3), keepdim=True)))))
    yr = d * g
    cot = _randn(tuple(yr.shape), 10)
    (yr * cot.double()).sum().backward()
    dw, r, e = dw.to(DEV), r.to(DEV), e.to(DEV)
    x = _nhwc(x0).requires_grad_()
    y = T.squeeze_excite(T.dwconv3x3(x, dw, pads), r, e)
    (y * cot.to(DEV)).sum().backward()
    assert rel_err(y, yr.detach()) < 2e-5
    assert rel_err(x.grad, xr.grad) < 1e-4
    for m_, mr in zip((dw, r, e), mods_r):
        for (n, p_), (_, pr) in zip(m_.named_parameters(), mr.named_parameters()):
            assert rel_err(p_.grad, pr.grad) < 1e-4, (type(m_).__name__, n)


def test_stem_weight_gradient_and_residual_join():
    torch.manual_seed(7)
    conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
    x0 = _randn((3, 3, 36, 52), 11)
    cr = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).double()
    cr.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    a0 = _randn((3, 64, 18, 26), 12)
    ar = a0.double().requires_grad_()
    yr = F.relu(cr(x0.double()) + ar)
    cot = _randn(tuple(yr.shape), 13)
    (yr * cot.double()).sum().backward()
    conv = conv.to(DEV)
    a = _nhwc(a0).requires_grad_()
    y = T.add(T.stem7x7(x0.to(DEV), conv), a, act=0.0)
    (y * cot.to(DEV)).sum().backward()
    assert rel_err(y, yr.detach()) < 2e-5
    assert rel_err(conv.weight.grad, cr.weight.grad) < 1e-4
    assert rel_err(a.grad, ar.grad) < 1e-6


# ------------------------------------------------------------------------------------------- matching encoder ---
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_matching_encoder_gradients_match_reference_autograd(mode):
    case = gc.MATCHING_CASES["small"]
    gold = np.load(os.path.join(GOLDEN, f"grad_matching_encoder_{mode}.npz"))
    enc = networks.ResnetMatchingEncoder(18, 16)
    synthetic.seeded_fill_(enc, seed=case["seed"])
    enc = enc.to(DEV)
    enc.train(mode == "train")
    x = gc.matching_input(case).to(DEV)
    y = enc(x)
    assert y.requires_grad
    cot = torch.from_numpy(gc.encoder_cotangent(case, tuple(y.shape))).to(DEV)
    (y * cot).sum().backward()
    assert rel_err(y, gold["out"]) < 1e-4, "forward (BatchNorm " + mode + ")"
    worst = {}
    # gradients below 1e-3 of the typical magnitude (the two biases in front of an InstanceNorm: exactly 0) are compared on that absolute scale
    floor = 1e-3 * float(np.median([np.sqrt((gold["d_" + n].astype(np.float64) ** 2).mean()) for n, _ in enc.named_parameters()]))
    for name, p in enc.named_parameters():
        assert p.grad is not None, name
        worst[name] = rel_l2(p.grad, gold["d_" + name], floor=floor)
    bad = {k: v for k, v in worst.items() if v > 5e-3}
    assert not bad, bad
    assert np.median(list(worst.values())) < 1e-3, worst
    if mode == "train":
        for name, b in enc.named_buffers():
            if "running" in name:
                assert rel_err(b, gold["buf_" + name]) < 1e-5, name
    # the inference path is untouched: eval + no_grad still runs the fused kernels and equals the training graph's
    # forward in eval mode
    if mode == "eval":
        with torch.inference_mode():
            yi = enc(x)
        assert rel_err(yi, y.detach()) < 1e-5


def test_matching_encoder_pair_in_training_mode_is_one_batch():
    """forward_pair under training = the reference's TensorFormatter call: ONE batch of B(1+K) images (BatchNorm
    statistics over all of them, depth_model.py:234-240)."""
    enc = networks.ResnetMatchingEncoder(18, 16)
    synthetic.seeded_fill_(enc, seed=3)
    enc = enc.to(DEV).train()
    cur, src = _randn((2, 3, 32, 48), 20).to(DEV), _randn((2, 3, 3, 32, 48), 21).to(DEV)
    fc, fs = enc.forward_pair(cur, src)
    sd0 = {k: v.clone() for k, v in enc.state_dict().items()}
    synthetic.seeded_fill_(enc, seed=3)
    allf = enc(torch.cat([cur.unsqueeze(1), src], 1).flatten(0, 1)).unflatten(0, (2, 4))
    assert torch.equal(fc, allf[:, 0]) and torch.equal(fs, allf[:, 1:]) and fc.requires_grad
    assert all(torch.equal(v, enc.state_dict()[k]) for k, v in sd0.items() if "num_batches" not in k)


# ------------------------------------------------------------------------------------------- image-prior encoder -
@pytest.mark.parametrize("training", [True, False])
def test_image_prior_encoder_gradients_match_aten_restatement(training):
    torch.manual_seed(0)
    enc = image_encoder.EfficientNetV2SFeatures()
    synthetic.seeded_fill_(enc, seed=6, gain=1.0)
    B, H, W = 2, 64, 96
    img = _randn((B, 3, H, W), 30)
    sd = {k: v.detach().clone().double().requires_grad_(v.is_floating_point() and "running" not in k)
          for k, v in enc.state_dict().items()}
    feats_r = effnet_torch.features(img.double(), sd, training=training)
    cots = [_randn(tuple(f.shape), 31 + i) for i, f in enumerate(feats_r)]
    sum((f * c.double()).sum() for f, c in zip(feats_r, cots)).backward()
    enc = enc.to(DEV)
    enc.train(training)
    feats = enc(img.to(DEV))
    assert all(f.requires_grad for f in feats)
    sum((f * c.to(DEV)).sum() for f, c in zip(feats, cots)).backward()
    for i, (f, fr) in enumerate(zip(feats, feats_r)):
        assert rel_err(f, fr.detach()) < 1e-4, f"pyramid level {i}"
    # relative L2 per parameter; parameters whose true gradient is (numerically) zero -- BatchNorm biases in front of a 1x1
    # conv + training-mode BatchNorm -- are held to an absolute bound: error RMS below 1e-4 of the typical gradient RMS
    scale = float(np.median([float(sd[n].grad.pow(2).mean().sqrt()) for n, _ in enc.named_parameters()]))
    worst = {n: rel_l2(p.grad, sd[n].grad, floor=1e-3 * scale) for n, p in enc.named_parameters()}
    bad = {k: v for k, v in worst.items() if v > 5e-3 and
           float((dict(enc.named_parameters())[k].grad.cpu().double() - sd[k].grad).pow(2).mean().sqrt()) > 1e-4 * scale}
    assert not bad, bad
    assert np.median(list(worst.values())) < 5e-4, np.median(list(worst.values()))


# ------------------------------------------------------------------------------------------- whole model ---------
def test_depth_model_trains_end_to_end_and_freeze_is_opt_in():
    B, K, H, W, D = 1, 2, 64, 96, 8
    opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=D)
    model = dm.DepthModel(opts)
    synthetic.seeded_fill_(model.encoder, seed=6, gain=1.0)
    for i, m in enumerate((model.matching_model, model.cost_volume_net, model.depth_decoder, model.cost_volume.mlp)):
        synthetic.seeded_fill_(m, seed=20 + i)
    model = model.to(DEV).train()
    inp = synthetic.cost_volume_inputs(B, K, 16, H // 4, W // 4, seed=4, device=DEV)
    eye = torch.eye(4, device=DEV).expand(B, 4, 4).contiguous()
    cur = {"image_b3hw": _randn((B, 3, H, W), 40).to(DEV), "invK_s1_b44": inp["cur_invK"], "cam_T_world_b44": eye,
           "world_T_cam_b44": eye}
    src = {"image_b3hw": _randn((B, K, 3, H, W), 41).to(DEV), "K_s1_b44": inp["src_Ks"],
           "cam_T_world_b44": inp["src_extrinsics"], "world_T_cam_b44": inp["src_poses"]}
    torch.manual_seed(1)   # phase "train" draws the flip augmentation
    out = model("train", cur, src)
    loss = sum(out[f"log_depth_pred_s{i}_b1hw"].abs().mean() for i in range(4))
    loss.backward()
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    # parameters the reference's graph does not reach either: heads of the deeper UNet++ columns that are overwritten
    assert all("depth_decoder" in n for n in missing), missing
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
    enc_grads = [p.grad for p in list(model.encoder.parameters()) + list(model.matching_model.parameters())]
    assert all(g is not None for g in enc_grads) and any(float(g.abs().max()) > 0 for g in enc_grads)
    # explicit opt-in: frozen encoders
    model.zero_grad(set_to_none=True)
    model.freeze_encoders = True
    out = model("train", cur, src)
    sum(out[f"log_depth_pred_s{i}_b1hw"].abs().mean() for i in range(4)).backward()
    assert all(p.grad is None for p in model.encoder.parameters())
    assert all(p.grad is None for p in model.matching_model.parameters())
    assert any(p.grad is not None for p in model.cost_volume_net.parameters())


def test_training_step_under_autocast_runs_the_fp32_kernels(monkeypatch):
    """The reference trains under 16-bit autocast (options.py:100-101, train.py:132).  Inside torch.autocast every
    differentiable HIP operator runs its fp32 kernels (half-precision inputs are upcast at the operator's entry): same
    outputs as without autocast, fp32 results, GradScaler-compatible.  What autograd SAVES for the backward pass is kept
    in the autocast dtype (autograd_ops._stash: the bulk of a training step's memory): less than 60 % of the fp32 bytes,
    gradients within half-precision rounding of the fp32 ones; SR_AUTOCAST_HALF_STORAGE=0 keeps fp32 storage and then the
    gradients equal the no-autocast ones up to summation order."""
    from simplerecon_amd import autograd_ops
    B, K, H, W, D = 1, 2, 64, 96, 8
    opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=D)
    model = dm.DepthModel(opts)
    synthetic.seeded_fill_(model.encoder, seed=6, gain=1.0)
    for i, m in enumerate((model.matching_model, model.cost_volume_net, model.depth_decoder, model.cost_volume.mlp)):
        synthetic.seeded_fill_(m, seed=20 + i)
    model = model.to(DEV).eval()   # eval-mode BatchNorm: identical forward passes
    inp = synthetic.cost_volume_inputs(B, K, 16, H // 4, W // 4, seed=4, device=DEV)
    eye = torch.eye(4, device=DEV).expand(B, 4, 4).contiguous()
    cur = {"image_b3hw": _randn((B, 3, H, W), 40).to(DEV), "invK_s1_b44": inp["cur_invK"], "cam_T_world_b44": eye,
           "world_T_cam_b44": eye}
    src = {"image_b3hw": _randn((B, K, 3, H, W), 41).to(DEV), "K_s1_b44": inp["src_Ks"],
           "cam_T_world_b44": inp["src_extrinsics"], "world_T_cam_b44": inp["src_poses"]}

    def step(autocast, half_images=False):
        model.zero_grad(set_to_none=True)
        c, s = dict(cur), dict(src)
        if half_images:
            c["image_b3hw"], s["image_b3hw"] = c["image_b3hw"].half(), s["image_b3hw"].half()
        saved = [0]

        def pack(t):
            if t.dim() == 4 and not isinstance(t, torch.nn.Parameter):
                saved[0] += t.numel() * t.element_size()
            return t
        with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
            with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
                out = model("val", c, s)
                loss = sum(out[f"log_depth_pred_s{i}_b1hw"].abs().mean() for i in range(4))
        scaler = torch.amp.GradScaler("cuda", enabled=autocast)
        scaler.scale(loss).backward()
        scale = float(scaler.get_scale()) if autocast else 1.0
        return out, {n: p.grad.clone() / scale for n, p in model.named_parameters() if p.grad is not None}, saved[0]

    ref_out, ref_g, bytes_fp32 = step(False)
    scale = float(np.median([float(v.pow(2).mean().sqrt()) for v in ref_g.values()]))
    # (a) + (b): the fp32-kernel mode (SR_AUTOCAST_HALF_IO=0, the r03 behaviour); 16-bit kernel I/O (the r04 default) is (c)
    monkeypatch.setattr(autograd_ops, "HALF_IO", False)
    # (a) fp32 storage under autocast: the same numbers
    monkeypatch.setattr(autograd_ops, "STORE_HALF", False)
    out, g, bytes_a = step(True)
    assert out["depth_pred_s0_b1hw"].dtype == torch.float32
    assert torch.equal(out["depth_pred_s0_b1hw"], ref_out["depth_pred_s0_b1hw"]) and bytes_a == bytes_fp32
    assert sorted(g) == sorted(ref_g)
    for n in g:   # (scaled-loss gradients differ from the unscaled run by fp32 rounding; zero gradients by noise)
        assert torch.isfinite(g[n]).all() and rel_l2(g[n], ref_g[n], floor=1e-3 * scale) < 1e-3, n
    # (b) 16-bit storage of the saved activations (the default): same forward, less memory, gradients within fp16 rounding
    monkeypatch.setattr(autograd_ops, "STORE_HALF", True)
    out, g, bytes_b = step(True)
    assert torch.equal(out["depth_pred_s0_b1hw"], ref_out["depth_pred_s0_b1hw"])
    assert bytes_b < 0.6 * bytes_fp32, (bytes_b, bytes_fp32)
    errs = {n: rel_l2(g[n], ref_g[n], floor=1e-2 * scale) for n in g}
    assert all(torch.isfinite(v).all() for v in g.values()) and max(errs.values()) < 5e-2 and np.median(list(errs.values())) < 5e-3, \
        (max(errs.values()), np.median(list(errs.values())))
    # (c) 16-bit kernel I/O in the conv stack (default): the forward now carries fp16 activations between the layers of
    # CVEncoder / DepthDecoderPP -- half-precision agreement with the fp32 run, every gradient finite
    monkeypatch.setattr(autograd_ops, "HALF_IO", True)
    out_c, g_c, bytes_c = step(True)
    assert out_c["depth_pred_s0_b1hw"].dtype == torch.float32 and torch.isfinite(out_c["depth_pred_s0_b1hw"]).all()
    assert rel_err(out_c["depth_pred_s0_b1hw"], ref_out["depth_pred_s0_b1hw"]) < 5e-2 and bytes_c < 0.6 * bytes_fp32
    errs_c = {n: rel_l2(g_c[n], ref_g[n], floor=1e-2 * scale) for n in g_c}
    assert sorted(g_c) == sorted(ref_g) and all(torch.isfinite(v).all() for v in g_c.values())
    assert np.median(list(errs_c.values())) < 5e-2, (max(errs_c.values()), np.median(list(errs_c.values())))
    monkeypatch.setattr(autograd_ops, "HALF_IO", False)
    # half-precision images (a caller that casts its batch): upcast at the first operator, fp32 from there on
    out_h, _, _ = step(True, half_images=True)
    assert out_h["depth_pred_s0_b1hw"].dtype == torch.float32 and torch.isfinite(out_h["depth_pred_s0_b1hw"]).all()
    assert rel_err(out_h["depth_pred_s0_b1hw"], ref_out["depth_pred_s0_b1hw"]) < 5e-2   # fp16-rounded inputs
