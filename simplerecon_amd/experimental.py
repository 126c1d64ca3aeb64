"""Experiment switches of the HIP library, as context managers.  Nothing here is on a default path.

`split_precision("f16")` turns on the two fenced split-precision kernels for the calls made inside
the block: the metadata-MLP sweep's layers 1-2 and the Winograd 3x3 convolutions then multiply on the 16-bit matrix pipe --
every fp32 operand as two 16-bit pieces, three exact products, fp32 accumulate.  Tensors, parameters and results stay fp32.
The library reads the switches per call (environment `SR_MLP_SPLIT`, `SR_WINO_SPLIT`), so the block must enclose the
forward calls themselves; packed weights are re-packed when the mode changes (`ops.packed_wino_weight`'s cache key)."""
import contextlib
import os

_MODES = ("f16", "bf16")


@contextlib.contextmanager
def split_precision(mode="f16", sweep=True, convs=True):
    """mode: "f16" (as close to fp64 as the fp32 kernels on everything measured; operands must stay inside fp16's range,
    |x| < 65504, or the result is non-finite) or "bf16" (2-25 x coarser, fp32's range).  `sweep` / `convs` select the
    kernels.  Restores the previous switches on exit."""
    if mode not in _MODES:
        raise ValueError(f"split_precision mode must be one of {_MODES}, got {mode!r}")
    names = [n for n, on in (("SR_MLP_SPLIT", sweep), ("SR_WINO_SPLIT", convs)) if on]
    saved = {n: os.environ.get(n) for n in names}
    try:
        for n in names:
            os.environ[n] = mode
        yield
    finally:
        for n, v in saved.items():
            if v is None:
                os.environ.pop(n, None)
            else:
                os.environ[n] = v
