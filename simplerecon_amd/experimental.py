"""Experiment switches of the HIP library, as context managers.  Nothing here is on a default path.

`split_precision("f16")` turns on the two fenced split-precision kernels for the calls made inside
the block: the metadata-MLP sweep's layers 1-2 and the Winograd 3x3 convolutions then multiply on the 16-bit matrix pipe --
every fp32 operand as two 16-bit pieces, three exact products, fp32 accumulate.  Tensors, parameters and results stay fp32.
The modes are entries of the library's option table (`SR_MLP_SPLIT`, `SR_WINO_SPLIT`: include/simplerecon_hip.h), set
through `_lib.set_option` -- the process environment is neither read nor written (r04 mutated os.environ, which the library
re-read with getenv() on every launch: a data race against launches on other threads, and inherited by child processes).
The table is process-wide: launches from OTHER threads inside the block run split as well, and a HIP graph captured inside
the block keeps the split kernels after it.  The block must enclose the forward calls themselves; packed weights are
re-packed when the mode changes (`ops.packed_wino_weight`'s cache key)."""
import contextlib

from . import _lib

_MODES = ("f16", "bf16")


@contextlib.contextmanager
def split_precision(mode="f16", sweep=True, convs=True):
    """mode: "f16" (as close to fp64 as the fp32 kernels on everything measured; operands must stay inside fp16's range,
    |x| < 65504, or the result is non-finite) or "bf16" (2-25 x coarser, fp32's range).  `sweep` / `convs` select the
    kernels.  Restores the previous switches on exit."""
    if mode not in _MODES:
        raise ValueError(f"split_precision mode must be one of {_MODES}, got {mode!r}")
    names = [n for n, on in (("SR_MLP_SPLIT", sweep), ("SR_WINO_SPLIT", convs)) if on]
    saved = {}
    try:
        for n in names:
            saved[n] = _lib.set_option(n, mode)
        yield
    finally:
        for n, v in saved.items():
            _lib.set_option(n, v)
