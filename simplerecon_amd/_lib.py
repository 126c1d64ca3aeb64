"""ctypes binding of libsimplerecon_hip.so (the C ABI of include/simplerecon_hip.h).

There is NO fallback: if the library is missing, was built for another architecture, or
the tensors are not fp32 device tensors, the call fails loudly."""
import contextlib
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SR_HIP_LIBRARY") or os.path.join(_HERE, "libsimplerecon_hip.so")   # override: ablation / trace builds
ABI_VERSION = 1

_lib = None

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/simplerecon_hip.h
SIGNATURES = {
    "sr_abi_version": (_i, []),
    "sr_option_count": (_i, []),
    "sr_option_name": (C.c_char_p, [_i]),
    "sr_option_id": (_i, [C.c_char_p]),
    "sr_option_get": (_i, [_i, C.POINTER(C.c_int)]),
    "sr_option_set": (_i, [_i, _i, C.POINTER(C.c_int)]),
    "sr_option_default": (_i, [_i, C.POINTER(C.c_int)]),
    "sr_target_arch": (C.c_char_p, []),
    "sr_volume_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sr_volume_prepare": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "sr_dot_volume_sweep": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i,
                                 _p, _i64, _i64, _i64, _p, _p, _p, _sz, _p]),
    "sr_dot_volume_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i,
                               _p, _i64, _i64, _i64, _p, _p, _p, _sz, _p]),
    "sr_gemm1x1_workspace_bytes": (_sz, []),
    "sr_gemm1x1_nhwc_fwd": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "sr_mbconv_fused_supported": (_i, [_i, _i, _i, _i, _i]),
    "sr_mbconv_expand_dw_se_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _p, _p, _i, _i, _i, _i, _i,
                                        _i, _p]),
    "sr_rgb_stem3x3s2_fwd": (_i, [_p, _i64, _i64, _i64, _i64, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "sr_se_gate2_fwd": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "sr_conv3x3_wino_io_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "sr_pw_conv_io_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _f, _i, _p]),
    "sr_pw_conv_tiled_workspace_bytes": (_sz, [_i, _i, _i]),
    "sr_pw_conv_tiled_plan": (_i, [_i, _i, _i, _i, _p, _p]),
    "sr_pw_conv_tiled_nhwc_fwd": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _f, _p, _sz, _p]),
    "sr_pw_conv_supported": (_i, [_i, _i]),
    "sr_pw_conv_plan": (_i, [_i, _i, _i, _i, _p, _p]),
    "sr_pw_conv_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _f, _p]),
    "sr_backproject_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "sr_project3d_fwd": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "sr_backproject_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "sr_project3d_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "sr_pose_distance_fwd": (_i, [_p, _p, _i, _p]),
    "sr_camera_rays_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "sr_selftest_rcp": (_i, [_p, _p, _p, _i, _p]),
    "sr_warp_features_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i,
                                  _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sr_mlp_volume_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "sr_mlp_pack_weights": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "sr_mlp_volume_sweep": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _f, _i, _i, _i, _i, _i, _i,
                                 _p, _i64, _i64, _i64, _p, _p, _p, _sz, _p]),
    "sr_mlp_volume_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _i,
                               _f, _i, _i, _i, _i, _i, _i, _p, _i64, _i64, _i64, _p, _p, _p, _sz, _p]),
    "sr_conv_packed_weight_floats": (_sz, [_i, _i, _i]),
    "sr_conv_pack_weights": (_i, [_p, _i, _i, _i, _p, _p]),
    "sr_conv2d_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "sr_conv2d_replicate_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i,
                                          _f, _p]),
    "sr_conv2d_padded_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i,
                                       _i, _i, _i, _i, _f, _p]),
    "sr_dwconv3x3_pool_bands": (_i, [_i]),
    "sr_dwconv3x3_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f,
                                   _p]),
    "sr_add_nhwc_fwd": (_i, [_p, _i64, _i, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "sr_se_scale_nhwc_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _i64, _i, _p, _i, _i, _i, _i, _i,
                                  _p]),
    "sr_se_gate_fwd": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "sr_scale_channels_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "sr_dot_volume_bwd_scratch_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sr_dot_volume_bwd": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _p, _p,
                               _p, _sz, _p, _sz, _p]),
    "sr_mlp_volume_bwd_scratch_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "sr_mlp_volume_bwd": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _f, _i, _i,
                               _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p, _sz, _p]),
    "sr_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "sr_conv_wgrad_nhwc": (_i, [_p, _i64, _i, _p, _i64, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "sr_bias_grad_nhwc": (_i, [_p, _i64, _i, _p, _i, _i, _i, _i, _p]),
    "sr_act_bwd": (_i, [_p, _p, _p, _i64, _f, _p]),
    "sr_act_bwd_bias_workspace_bytes": (_sz, [_i64, _i]),
    "sr_act_bwd_bias_nhwc": (_i, [_p, _p, _p, _p, _i64, _i, _f, _p, _sz, _p]),
    "sr_zero_stuff2x_nhwc": (_i, [_p, _i64, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sr_upsample2x_bwd_nhwc": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "sr_conv_flip_transpose_weights": (_i, [_p, _i, _i, _i, _p, _p]),
    "sr_mul_fwd": (_i, [_p, _p, _p, _i64, _p]),
    "sr_norm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sr_norm_stats_nhwc": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "sr_norm_act_fwd_nhwc": (_i, [_p, _i64, _i, _p, _p, _f, _p, _p, _f, _i, _p, _i64, _i, _i, _i, _i, _p]),
    "sr_norm_act_bwd_nhwc": (_i, [_p, _i64, _i, _p, _i64, _i, _p, _p, _f, _p, _p, _f, _i, _i, _p, _i64, _i, _p, _p,
                                  _i, _i, _i, _p, _sz, _p]),
    "sr_rowsum_nhwc": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _i, _i, _f, _p, _p, _sz, _p]),
    "sr_maxblurpool_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sr_maxblurpool_bwd_nhwc": (_i, [_p, _i64, _i, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "sr_replicate_pad_nhwc_fwd": (_i, [_p, _i64, _i, _p, _i, _i, _i, _i, _i, _p]),
    "sr_replicate_pad_nhwc_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "sr_im2col7x7s2_nhwc": (_i, [_p, _i64, _i64, _i64, _i64, _p, _i, _i, _i, _i, _p]),
    "sr_dwconv3x3_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sr_dwconv3x3_bwd_nhwc": (_i, [_p, _i64, _i, _p, _i64, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz,
                                   _p]),
    "sr_scale_bwd_nhwc": (_i, [_p, _i64, _i, _p, _p, _f, _p, _i, _i, _i, _p]),
    "sr_small_linear_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "sr_small_linear_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "sr_act_in_bwd": (_i, [_p, _p, _p, _i64, _f, _p]),
    "sr_add_act_fwd": (_i, [_p, _p, _p, _p, _i64, _f, _p]),
    "sr_conv_wgrad_padded_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "sr_conv_wgrad_padded_nhwc": (_i, [_p, _i64, _i, _p, _i64, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p,
                                       _sz, _p]),
    "sr_conv_splitk_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "sr_conv2d_splitk_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i,
                                       _f, _p, _sz, _p]),
    "sr_wino_packed_weight_floats": (_sz, [_i, _i]),
    "sr_wino_pack_weights": (_i, [_p, _i, _i, _p, _p]),
    "sr_conv_prefers_wino": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "sr_wino_splitk_factor": (_i, [_i, _i, _i, _i, _i]),
    "sr_wino_splitk_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sr_conv3x3_wino_splitk_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _f,
                                             _p, _sz, _p]),
    "sr_wino_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i]),
    "sr_conv3x3_wino_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _f, _p]),
    "sr_wino4_packed_weight_floats": (_sz, [_i, _i]),
    "sr_wino4_pack_weights": (_i, [_p, _i, _i, _p, _p]),
    "sr_conv_prefers_wino4": (_i, [_i, _i, _i, _i, _i, _i]),
    "sr_conv3x3_wino4_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _f, _p]),
    "sr_conv3x3_wino4_variant_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _f, _i,
                                               _p]),
    "sr_conv_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "sr_upsample2x_nhwc_fwd": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "sr_exp_fwd": (_i, [_p, _p, _i64, _p]),
    "sr_stem_packed_weight_floats": (_sz, [_i]),
    "sr_stem_pack_weights": (_i, [_p, _i, _p, _p]),
    "sr_stem7x7_fwd": (_i, [_p, _i64, _i64, _i64, _i64, _p, _p, _p, _f, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "sr_maxblurpool_nhwc_fwd": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "sr_instance_norm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sr_instance_norm_nhwc_fwd": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _i, _i, _i, _f, _f, _p, _sz, _p]),
    "sr_instance_norm_stats_nhwc": (_i, [_p, _i64, _i, _i, _i, _i, _i, _f, _p, _p, _sz, _p]),
    "sr_conv1x1_stats_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sr_conv1x1_stats_nhwc_fwd": (_i, [_p, _i64, _i, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _f, _p, _p, _sz, _p]),
    "sr_conv3x3_c16_packed_weight_floats": (_sz, [_i, _i]),
    "sr_conv3x3_c16_pack_weights": (_i, [_p, _i, _i, _p, _p]),
    "sr_conv3x3_c16_nhwc_fwd": (_i, [_p, _i64, _i, _p, _f, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "sr_tsdf_integrate_fwd": (_i, [_p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _i, _i, _i,
                                   _f, _f, _f, _f, _f, _p]),
}


class HipLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -m simplerecon_amd.build` "
                "(hipcc --offload-arch=gfx950).  simplerecon_amd has no CPU/torch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if l.sr_abi_version() != ABI_VERSION:
            raise HipLibraryError(f"ABI mismatch: library {l.sr_abi_version()} vs binding {ABI_VERSION}")
        _lib = l
    return _lib


_ERRORS = {1: "invalid argument", 2: "unsupported configuration", 3: "workspace too small"}


def check(rc, what):
    if rc != 0:
        msg = _ERRORS.get(rc, f"hipError {rc - 1000}" if rc >= 1000 else f"error {rc}")
        raise HipLibraryError(f"{what} failed: {msg}")


# ---- run-time switches: the library's option table (include/simplerecon_hip.h, SR_OPT_*; r05) --------------------------
# Options are named like the environment variable that seeds them at the library's first access ("SR_WINO_XCD"); after that
# the library never reads the environment again and hosts change a switch through these calls.  Process-wide.
SPLIT_MODES = {"": 0, "0": 0, "off": 0, "fp32": 0, "bf16": 1, "f16": 2, "fp16": 2}
_SPLIT_NAMES = {0: "", 1: "bf16", 2: "f16"}
SPLIT_OPTION_NAMES = ("SR_MLP_SPLIT", "SR_WINO_SPLIT")   # the options that take a mode NAME; every other option is an integer
# The split modes are read by the weight-packing AND by the launching entry points: a change between the two would feed one
# format's packed weights to the other format's kernel.  Changing them and [pack + launch] hold this lock.
import threading  # noqa: E402
SPLIT_GUARD = threading.RLock()


OPTION_LISTENERS = []   # callables (name, value) run after every set_option: caches of plan queries hang themselves in here


def _option_id(name):
    oid = lib().sr_option_id(name.encode())
    if oid < 0:
        raise KeyError(f"libsimplerecon_hip.so has no option {name!r}")
    return oid


_OPTION_MIRROR = {}   # name -> value as last read / set THROUGH this module: what the launch paths consult (no ctypes call
                      # per launch).  A host that drives sr_option_set() by other means calls refresh_options() afterwards.


def get_option(name):
    v = _OPTION_MIRROR.get(name)
    if v is None:
        c = C.c_int(0)
        check(lib().sr_option_get(_option_id(name), C.byref(c)), "sr_option_get")
        v = _OPTION_MIRROR[name] = c.value
    return v


def refresh_options():
    _OPTION_MIRROR.clear()


def set_option(name, value):
    """Sets option `name` (an int; the split modes also take 'bf16' / 'f16' / '' / an unknown string = -1: refused by the
    entry points).  Returns the previous value."""
    if isinstance(value, str):
        if name not in SPLIT_OPTION_NAMES:
            raise ValueError(f"option {name} takes an integer, got {value!r}")   # (a typo used to become -1 silently)
        value = SPLIT_MODES.get(value, -1)
    prev = C.c_int(0)
    if name in SPLIT_OPTION_NAMES:
        with SPLIT_GUARD:
            check(lib().sr_option_set(_option_id(name), int(value), C.byref(prev)), "sr_option_set")
    else:
        check(lib().sr_option_set(_option_id(name), int(value), C.byref(prev)), "sr_option_set")
    _OPTION_MIRROR[name] = int(value)
    for fn in OPTION_LISTENERS:
        fn(name, int(value))
    return prev.value


@contextlib.contextmanager
def option(name, value):
    """`with _lib.option("SR_PW_NT", 2): ...` -- the switch for the calls inside the block, restored on exit."""
    prev = set_option(name, value)
    try:
        yield
    finally:
        set_option(name, prev)


def split_mode_name(option_name):
    """'' / 'bf16' / 'f16' for the split-precision options (anything else: the raw integer as a string)."""
    v = get_option(option_name)
    return _SPLIT_NAMES.get(v, str(v))


def require_device_f32(name, t, allow_none=False):
    if t is None:
        if allow_none:
            return
        raise ValueError(f"{name} is None")
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise HipLibraryError(f"{name} lives on {t.device}: the HIP path needs device tensors (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference runs inference in fp32), got {t.dtype}")


def ptr(t):
    """Device address of a tensor for a `void*` / `float*` argument (a plain int: ctypes converts it; None = NULL)."""
    return t.data_ptr() if t is not None else None


def stream_ptr(device=None):
    """torch's current HIP stream on `device` as the `void* stream` of the C ABI (the raw handle: no Stream object)."""
    idx = -1 if device is None or device.index is None else device.index
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def stream_id(device=None):
    """The same handle as a plain int (dictionary keys of per-stream caches)."""
    return stream_ptr(device).value or 0


_CUDA_AVAILABLE = None


def cuda_available():
    """torch.cuda.is_available(), asked once (it re-reads the environment on every call: ~1 us on the launch path)."""
    global _CUDA_AVAILABLE
    if _CUDA_AVAILABLE is None:
        _CUDA_AVAILABLE = torch.cuda.is_available()
    return _CUDA_AVAILABLE


def capturing():
    """True while torch's current stream is being captured into a HIP graph (False on a machine without a GPU)."""
    return cuda_available() and torch.cuda.is_current_stream_capturing()


_NULL_CONTEXT = contextlib.nullcontext()


def on_device(device):
    """Context that makes `device` the current HIP device around a launch -- nothing to do when it already is (the
    single-GPU-per-process case: one process per GPU is the deployment model)."""
    if device.type != "cuda" or device.index is None or device.index == torch.cuda.current_device():
        return _NULL_CONTEXT
    return torch.cuda.device(device)


def refuse_autograd(*tensors):
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "simplerecon_amd implements the inference hot path only; run under torch.inference_mode() / "
            "no_grad() (the backward pass is listed as a 'next' component, SURVEY.md §8f)")
