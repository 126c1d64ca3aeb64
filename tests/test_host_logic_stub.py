"""Host-side logic of the cost-volume managers and of the whole DepthModel on CPU: the C-ABI library is replaced by a stub that checks every call
against the ctypes signature table (argument count and kinds) and returns success, so the Python control flow --
argument packing, output allocation, mask handling, the autograd seams and their bookkeeping -- runs without a GPU.
No numerics here: the kernels themselves are covered by the `-m gpu` tests."""
import contextlib
import os
import ctypes

import pytest
import torch

from simplerecon_amd import _lib, synthetic
from simplerecon_amd import cost_volume as cv

B, K, C, H, W, D = 2, 3, 16, 8, 12, 4


@pytest.fixture
def stub(monkeypatch):
    class Calls(list):
        prefer_wino = False
    calls = Calls()

    class Stream:
        cuda_stream = 0

    class Fn:
        def __init__(self, name):
            self.name = name

        def __call__(self, *args):
            res, argtypes = _lib.SIGNATURES[self.name]
            assert len(args) == len(argtypes), (self.name, len(args), len(argtypes))
            for v, t in zip(args, argtypes):
                if t is ctypes.c_void_p:
                    assert v is None or isinstance(v, (ctypes.c_void_p, int)), (self.name, type(v))
                elif t is ctypes.c_float:
                    assert isinstance(v, (float, ctypes.c_float)), (self.name, type(v))
                else:
                    assert isinstance(v, int), (self.name, type(v), v)
            calls.append(self.name)
            if self.name == "sr_conv_prefers_wino":
                return int(calls.prefer_wino)
            if res is ctypes.c_char_p:
                return b"stub_kernel"
            return 4096 if res is ctypes.c_size_t else 0

    real = _lib.lib()   # (the option table is host-only code: those calls go to the real library, GPU or not)

    class Lib:
        def __getattr__(self, name):
            return getattr(real, name) if name.startswith("sr_option_") else Fn(name)

    from simplerecon_amd import ops
    monkeypatch.setattr(ops, "_SHAPE_QUERIES", {})   # per-shape answers of the (stubbed) library are cached per process
    monkeypatch.setattr(_lib, "lib", lambda: Lib())
    monkeypatch.setattr(_lib, "require_device_f32", lambda *a, **k: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda dev=None: ctypes.c_void_p(0))
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: Stream())
    return calls


@pytest.mark.parametrize("cls,kw,fwd", [(cv.CostVolumeManager, {}, "sr_dot_volume_fwd"),
                                        (cv.FeatureVolumeManager, dict(num_source_views=K), "sr_mlp_volume_fwd"),
                                        (cv.FastFeatureVolumeManager, dict(num_source_views=K), "sr_mlp_volume_fwd")])
def test_inference_paths(stub, cls, kw, fwd):
    inp = synthetic.cost_volume_inputs(B, K, C, H, W, seed=1)
    mgr = cls(H, W, num_depth_bins=D, **kw)
    for return_mask in (False, True):
        del stub[:]
        with torch.inference_mode():
            vol, lowest, planes, mask = mgr(return_mask=return_mask, **inp)
        assert vol.shape == (B, D, H, W) and lowest.shape == (B, H, W) and planes.shape == (B, D, H, W)
        # the dot model ignores return_mask (reference cost_volume.py:286, 335)
        assert (mask is None) == (not return_mask or cls is cv.CostVolumeManager)
        assert mask is None or mask.dtype == torch.bool
        assert stub[-1] == fwd
    # grad mode on: like any nn.Module, a manager whose parameters require grad builds an autograd graph (the MLP
    # managers -- reference train.py trains the matching MLP); with no parameters and constant features it does not
    assert mgr(**inp)[0].requires_grad == (cls is not cv.CostVolumeManager)
    for p_ in mgr.parameters():
        p_.requires_grad_(False)
    assert not mgr(**inp)[0].requires_grad
    # empty batch: nothing is launched
    empty = {k: (v[:0] if v.dim() > 0 and v.shape[0] == B and k not in ("min_depth", "max_depth") else v)
             for k, v in inp.items()}
    del stub[:]
    with torch.inference_mode():
        vol = mgr(**empty)[0]
    assert vol.shape == (0, D, H, W) and not [c for c in stub if c.endswith("_fwd")]


def test_autograd_seams(stub):
    inp = synthetic.cost_volume_inputs(B, K, C, H, W, seed=1)
    # dot model: differentiable w.r.t. the features out of the box
    dot = cv.CostVolumeManager(H, W, num_depth_bins=D)
    cur, src = inp["cur_feats"].clone().requires_grad_(), inp["src_feats"].clone().requires_grad_()
    vol, lowest, _, _ = dot(**dict(inp, cur_feats=cur, src_feats=src))
    assert vol.requires_grad and not lowest.requires_grad
    vol.sum().backward()
    assert stub[-2:] == ["sr_volume_prepare", "sr_dot_volume_bwd"] and cur.grad.shape == cur.shape
    assert src.grad.shape == src.shape
    with pytest.raises(NotImplementedError):
        dot(**dict(inp, src_Ks=inp["src_Ks"].clone().requires_grad_()))
    # MLP model: differentiable out of the box (a swapped-in manager trains under train.py); opting out refuses
    hero = cv.FeatureVolumeManager(H, W, num_depth_bins=D, num_source_views=K)
    hero.differentiable = False
    with pytest.raises(NotImplementedError):
        hero(**dict(inp, cur_feats=cur))
    del hero.differentiable
    assert hero.differentiable
    cur.grad = None
    vol, lowest, _, mask = hero(**dict(inp, cur_feats=cur), return_mask=True)
    assert vol.requires_grad and not lowest.requires_grad and mask.dtype == torch.bool and not mask.requires_grad
    vol.sum().backward()
    assert stub[-2:] == ["sr_volume_prepare", "sr_mlp_volume_bwd"]
    assert cur.grad is not None and all(p.grad is not None and p.grad.shape == p.shape for p in hero.mlp.parameters())
    with torch.no_grad():
        assert not hero(**dict(inp, cur_feats=cur))[0].requires_grad


@pytest.mark.parametrize("prefer_wino", [False, True])
@pytest.mark.parametrize("fvt", ["mlp_feature_volume", "simple_cost_volume"])
def test_whole_model_control_flow(stub, prefer_wino, fvt):
    """DepthModel.forward end to end through the stub: every stage reaches its entry point with a well-formed call and
    the output dict has the reference's keys and shapes (depth_model.py:388-405)."""
    from simplerecon_amd import depth_model as dm
    stub.prefer_wino = prefer_wino
    b, k, h, w, d = 1, 2, 96, 128, 8
    opts = dm.default_options(image_width=w, image_height=h, model_num_views=k + 1, matching_num_depth_bins=d,
                              feature_volume_type=fvt)
    model = dm.DepthModel(opts).eval()
    inp = synthetic.cost_volume_inputs(b, k, 16, h // 4, w // 4, seed=2)
    eye = torch.eye(4).expand(b, 4, 4).contiguous()
    cur = {"image_b3hw": torch.randn(b, 3, h, w), "invK_s1_b44": inp["cur_invK"], "cam_T_world_b44": eye,
           "world_T_cam_b44": eye}
    src = {"image_b3hw": torch.randn(b, k, 3, h, w), "K_s1_b44": inp["src_Ks"],
           "cam_T_world_b44": inp["src_extrinsics"], "world_T_cam_b44": inp["src_poses"]}
    with torch.inference_mode():
        out = model("test", cur, src, return_mask=True)
    for i in range(4):
        assert out[f"log_depth_pred_s{i}_b1hw"].shape == (b, 1, (h // 2) >> i, (w // 2) >> i)
        assert out[f"depth_pred_s{i}_b1hw"].shape == (b, 1, (h // 2) >> i, (w // 2) >> i)
    assert out["lowest_cost_bhw"].shape == (b, h // 4, w // 4)
    assert (out["overall_mask_bhw"] is None) == (fvt == "simple_cost_volume")
    seen = set(stub)
    sweep = "sr_mlp_volume_fwd" if fvt == "mlp_feature_volume" else "sr_dot_volume_fwd"
    for name in (sweep, "sr_stem7x7_fwd", "sr_maxblurpool_nhwc_fwd", "sr_conv1x1_stats_nhwc_fwd", "sr_conv3x3_c16_nhwc_fwd",
                 "sr_conv2d_padded_nhwc_fwd", "sr_dwconv3x3_nhwc_fwd", "sr_se_gate2_fwd", "sr_pw_conv_nhwc_fwd", "sr_add_nhwc_fwd",
                 "sr_upsample2x_nhwc_fwd", "sr_exp_fwd",
                 "sr_conv3x3_wino_splitk_nhwc_fwd" if prefer_wino else "sr_conv2d_splitk_nhwc_fwd"):
        assert name in seen, name
    assert stub.count("sr_dwconv3x3_nhwc_fwd") == 30 and stub.count("sr_upsample2x_nhwc_fwd") == 16


def test_autocast_region_upcasts_half_features(stub):
    """Reference training runs under 16-bit autocast (options.py:100-101): inside an autocast region half-precision
    matching features are upcast to fp32 for the HIP kernels and the gradients come back in the caller's dtype."""
    inp = synthetic.cost_volume_inputs(B, K, C, H, W, seed=1)
    dot = cv.CostVolumeManager(H, W, num_depth_bins=D)
    cur = inp["cur_feats"].bfloat16().requires_grad_()
    src = inp["src_feats"].bfloat16().requires_grad_()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        vol = dot(**dict(inp, cur_feats=cur, src_feats=src))[0]
    assert vol.dtype == torch.float32 and vol.requires_grad
    vol.sum().backward()
    assert cur.grad.dtype == torch.bfloat16 and src.grad.dtype == torch.bfloat16
    assert stub[-1] == "sr_dot_volume_bwd"


def test_packed_weight_cache_key_follows_the_parameters():
    """ops._state_key decides whether a cached packed weight is still valid: it must change when a parameter is updated in
    place (optimizer step, load_state_dict), replaced by a new tensor (module.to(...), re-assignment), or when the folded
    BatchNorm's statistics move -- and must not change otherwise."""
    from torch import nn

    from simplerecon_amd import ops
    conv, bn = nn.Conv2d(8, 16, 3, padding=1), nn.BatchNorm2d(16)
    k0, k0_bn = ops._state_key(conv, None), ops._state_key(conv, bn)
    assert ops._state_key(conv, None) == k0 and ops._state_key(conv, bn) == k0_bn and k0 != k0_bn
    with torch.no_grad():
        conv.weight.mul_(2.0)                                   # in-place update: version bump
    k1 = ops._state_key(conv, None)
    assert k1 != k0
    conv.load_state_dict({k: v.clone() for k, v in conv.state_dict().items()})   # copy_ into the same storage
    k2 = ops._state_key(conv, None)
    assert k2 != k1
    conv.weight = nn.Parameter(conv.weight.detach().clone())    # a new tensor object
    k3 = ops._state_key(conv, None)
    assert k3 != k2
    conv.bias = None
    assert ops._state_key(conv, None) != k3 and len(ops._state_key(conv, None)) == 2
    kb = ops._state_key(conv, bn)
    bn.running_var.add_(1.0)
    assert ops._state_key(conv, bn) != kb
    plain = nn.BatchNorm2d(16, affine=False)
    assert len(ops._state_key(conv, plain)) == 3               # weight + the two running statistics


def test_per_shape_library_answers_are_asked_once(stub):
    from simplerecon_amd import ops
    lib = _lib.lib()
    assert ops._shape_query(lib, "sr_conv_prefers_wino", 1, 8, 16, 16, 32, 3, 1) == 0
    n = len(stub)
    stub.prefer_wino = True     # the library's answer is a pure function of the shape: the cached one is returned
    assert ops._shape_query(lib, "sr_conv_prefers_wino", 1, 8, 16, 16, 32, 3, 1) == 0 and len(stub) == n
    assert ops._shape_query(lib, "sr_conv_prefers_wino", 2, 8, 16, 16, 32, 3, 1) == 1 and len(stub) == n + 1


def test_split_precision_switch_repacks_the_winograd_weight(stub, monkeypatch, sr_option):
    """SR_WINO_SPLIT (the fenced split-precision Winograd variant) changes what a packed weight CONTAINS (16-bit pieces in the
    fp32 layout's buffer): the cache must re-pack when the switch changes and hit otherwise; off-values mean the fp32 path."""
    from torch import nn

    from simplerecon_amd import ops
    monkeypatch.setattr(ops, "_await_packed", lambda *a, **k: None)
    monkeypatch.setattr(ops, "_packed_here", lambda *a, **k: None)
    conv = nn.Conv2d(16, 32, 3, padding=1)
    for off in ("", "0", "off", "fp32"):
        sr_option("SR_WINO_SPLIT", off)
        assert ops.wino_split_mode() == ""
    ops.packed_wino_weight(conv)
    n = stub.count("sr_wino_pack_weights")
    ops.packed_wino_weight(conv)
    assert stub.count("sr_wino_pack_weights") == n            # cache hit
    sr_option("SR_WINO_SPLIT", "f16")
    assert ops.wino_split_mode() == "f16"
    ops.packed_wino_weight(conv)
    assert stub.count("sr_wino_pack_weights") == n + 1        # re-packed as 16-bit pieces
    ops.packed_wino_weight(conv)
    assert stub.count("sr_wino_pack_weights") == n + 1
    sr_option("SR_WINO_SPLIT", "bf16")
    ops.packed_wino_weight(conv)
    assert stub.count("sr_wino_pack_weights") == n + 2
    sr_option("SR_WINO_SPLIT", "0")
    ops.packed_wino_weight(conv)
    assert stub.count("sr_wino_pack_weights") == n + 3        # and back to fp32 fragments


def test_fenced_workloads_label_their_arithmetic():
    """The split-precision workloads are separate registry entries whose bench line says what they compute with; the default
    workload is the fp32 one."""
    import bench_workloads as bw
    assert bw.DEFAULT == "hero_cfg3" and bw.HeroCfg3.dtype == "f32"
    for name in ("hero_cfg3_bf16x3", "hero_cfg3_f16x3", "hero_cfg3_bf16x3_convs", "hero_cfg3_f16x3_convs"):
        assert name in bw.WORKLOADS


def test_split_precision_context_manager_sets_and_restores_the_switches(sr_option):
    """The context manager drives the library's option table (no os.environ mutation: ADVICE r04) and nests."""
    from simplerecon_amd import experimental, ops
    env_before = dict(os.environ)
    sr_option("SR_MLP_SPLIT", 0)
    sr_option("SR_WINO_SPLIT", "bf16")
    mlp = lambda: _lib.split_mode_name("SR_MLP_SPLIT")
    with experimental.split_precision("f16"):
        assert mlp() == "f16" and ops.wino_split_mode() == "f16"
        with experimental.split_precision("bf16", convs=False):
            assert mlp() == "bf16" and ops.wino_split_mode() == "f16"
        assert mlp() == "f16"
    assert mlp() == "" and ops.wino_split_mode() == "bf16"
    assert dict(os.environ) == env_before
    with pytest.raises(ValueError):
        with experimental.split_precision("int8"):
            pass


def test_batchnorm_layer_cache_sees_replacements_deep_in_the_tree():
    """the cached list of batch-norm layers behind `any_batchnorm_training` must be rebuilt when a layer is
    swapped ANYWHERE in the tree (SyncBatchNorm conversion, fusion), and must recognise every `_BatchNorm` flavour."""
    from torch import nn
    from simplerecon_amd import autograd_ops as A
    m = nn.Sequential(nn.Sequential(nn.Conv2d(3, 3, 1), nn.BatchNorm2d(3)), nn.ReLU()).eval()
    assert not A.any_batchnorm_training(m)
    m[0][1] = nn.BatchNorm2d(3)                      # a fresh layer (training mode) two levels down
    assert A.any_batchnorm_training(m)
    m.eval()
    assert not A.any_batchnorm_training(m)
    m[0][1] = nn.SyncBatchNorm(3)                    # not an nn.BatchNorm2d, still batch statistics
    assert A.any_batchnorm_training(m)
    m[0][1].eval()
    assert not A.any_batchnorm_training(m)
    m[0][1].train()                                  # a flag flipped on the leaf only
    assert A.any_batchnorm_training(m)


def test_batchnorm_layer_cache_sees_direct_mutations_of_modules():
    """torch's registration hook does not fire for direct `_modules` mutations (del m.bn, ModuleList.__delitem__,
    fx / quantisation swaps); a cached layer that no longer sits where it was found invalidates the cache."""
    from torch import nn
    from simplerecon_amd import autograd_ops as A
    inner = nn.Sequential(nn.BatchNorm2d(3))
    m = nn.Sequential(nn.Conv2d(3, 3, 1), inner).eval()
    assert not A.any_batchnorm_training(m)
    fresh = nn.BatchNorm2d(3)                        # training mode
    inner._modules["0"] = fresh                      # swapped in behind the hook's back
    assert A.any_batchnorm_training(m)
    del inner._modules["0"]                          # and removed the same way
    assert not A.any_batchnorm_training(m)
    bn = nn.BatchNorm2d(2)                           # a batch-norm layer as the root of the query
    assert A.any_batchnorm_training(bn)
    assert not A.any_batchnorm_training(bn.eval())
