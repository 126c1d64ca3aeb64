"""Import shim for the UPSTREAM reference (test infrastructure only).

Used ONLY by tests/golden/make_golden.py, in the build container where
/root/reference exists, to run the reference's own PyTorch modules on CPU and
write golden vectors.  Nothing in the product path, bench.py or the GPU tests
imports this file, and /root/reference is never read at run time on the GPU box.

The reference's modules import third-party packages that are absent from this
image (kornia, torchvision, timm, antialiased_cnns) at module top, but the hot
path (modules/cost_volume.py, modules/networks.py {MLP, CVEncoder,
DepthDecoderPP}, modules/layers.py, utils/geometry_utils.py) never calls them.
We register empty stub modules so that the imports succeed.  `pyrdown`
(utils/generic_utils.py:87-94) is TorchScript-compiled at import and needs a
scriptable `kornia.filters.blur_pool2d`, so the stub provides real-source
functions.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SIMPLERECON_REFERENCE", "/root/reference")


def _stub_source():
    return '''
import torch
def blur_pool2d(input: torch.Tensor, kernel_size: int, stride: int = 2) -> torch.Tensor:
    return torch.nn.functional.avg_pool2d(input, kernel_size, stride, kernel_size // 2)
def gaussian_blur2d(input: torch.Tensor, kernel_size: tuple[int, int], sigma: tuple[float, float]) -> torch.Tensor:
    return input
def spatial_gradient(input: torch.Tensor) -> torch.Tensor:
    return torch.stack([input, input], dim=2)
'''


def _aa_blurpool_class():
    """torch.nn restatement of antialiased_cnns.BlurPool (adobe/antialiased-cnns, blurpool.py) for the one
    configuration the reference reaches: filt_size=4, stride=2, pad_type='reflect', pad_off=0.  The package
    itself is absent from this image and from /root/reference (simplerecon_env.yml:20, unpinned)."""
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class BlurPool(nn.Module):
        def __init__(self, channels, pad_type="reflect", filt_size=4, stride=2, pad_off=0):
            super().__init__()
            assert pad_type == "reflect" and pad_off == 0
            self.filt_size, self.stride, self.channels = filt_size, stride, channels
            lo, hi = int(1.0 * (filt_size - 1) / 2), int(np.ceil(1.0 * (filt_size - 1) / 2))
            self.pad = nn.ReflectionPad2d([lo, hi, lo, hi])
            a = {1: [1.0], 2: [1.0, 1.0], 3: [1.0, 2.0, 1.0], 4: [1.0, 3.0, 3.0, 1.0],
                 5: [1.0, 4.0, 6.0, 4.0, 1.0]}[filt_size]
            a = np.array(a)
            filt = torch.Tensor(a[:, None] * a[None, :])
            filt = filt / torch.sum(filt)
            self.register_buffer("filt", filt[None, None, :, :].repeat((channels, 1, 1, 1)))

        def forward(self, inp):
            return F.conv2d(self.pad(inp), self.filt, stride=self.stride, groups=inp.shape[1])

    return BlurPool


def _aa_resnet18(pretrained=False, filter_size=4, pool_only=True, **kwargs):
    """torch.nn restatement of the part of antialiased_cnns.resnet18 that ResnetMatchingEncoder keeps
    (modules/networks.py:176-182: conv1, bn1, relu, maxpool, layer1).  layer2..4 and fc are never used by
    the reference and are not built."""
    import torch.nn as nn
    if pretrained:
        raise RuntimeError("no network: pretrained antialiased_cnns weights are unavailable")
    BlurPool = _aa_blurpool_class()

    class Block(nn.Module):  # stride-1 BasicBlock of (antialiased) ResNet-18
        def __init__(self, planes):
            super().__init__()
            self.conv1 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)

        def forward(self, x):
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            out += x
            return self.relu(out)

    class Stem(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=1),
                                         BlurPool(64, filt_size=filter_size, stride=2))
            self.layer1 = nn.Sequential(Block(64), Block(64))

    return Stem()


def install_stubs():
    import tempfile
    if "kornia" in sys.modules and getattr(sys.modules["kornia"], "_sr_stub", False):
        return
    # kornia.filters must come from a real file so TorchScript can read source.
    d = tempfile.mkdtemp(prefix="sr_stub_")
    os.makedirs(os.path.join(d, "kornia"))
    with open(os.path.join(d, "kornia", "__init__.py"), "w") as f:
        f.write("from . import filters\n_sr_stub = True\n")
    with open(os.path.join(d, "kornia", "filters.py"), "w") as f:
        f.write(_stub_source())
    sys.path.insert(0, d)
    importlib.import_module("kornia")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "torchvision" not in sys.modules:
        tv = mod("torchvision")
        tv.models = mod("torchvision.models")
        tv.ops = mod("torchvision.ops", FeaturePyramidNetwork=object)
        tv.transforms = mod("torchvision.transforms")
        tv.transforms.functional = mod("torchvision.transforms.functional")
    if "timm" not in sys.modules:
        mod("timm")
    if "antialiased_cnns" not in sys.modules:
        mod("antialiased_cnns", resnet18=_aa_resnet18, resnet34=None, resnet50=None, resnet101=None,
            resnet152=None, BlurPool=_aa_blurpool_class())
    if "pytorch_lightning" not in sys.modules:
        pass  # DepthModel itself is not importable; not needed for the hot path


def import_keyframe_buffer():
    """The reference's tools/keyframe_buffer.py (numpy only)."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module("tools.keyframe_buffer")


def import_tsdf():
    """The reference's tools/tsdf.py (TSDF, TSDFFuser) with stubs for its mesh-export imports (trimesh, skimage),
    which integrate_depth never touches."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    for name in ("trimesh", "skimage", "skimage.measure"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    if not hasattr(sys.modules["trimesh"], "Trimesh"):
        sys.modules["trimesh"].Trimesh = object
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module("tools.tsdf")


def import_reference():
    """Returns the reference's (cost_volume, networks, layers, geometry_utils,
    generic_utils) modules imported unmodified from REFERENCE_ROOT."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    cv = importlib.import_module("modules.cost_volume")
    nets = importlib.import_module("modules.networks")
    layers = importlib.import_module("modules.layers")
    geo = importlib.import_module("utils.geometry_utils")
    gen = importlib.import_module("utils.generic_utils")
    return cv, nets, layers, geo, gen
