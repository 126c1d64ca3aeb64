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
        mod("antialiased_cnns")
    if "pytorch_lightning" not in sys.modules:
        pass  # DepthModel itself is not importable; not needed for the hot path


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
