"""Parity metrics.  The bar (BASELINE.json north_star): 1e-4 relative, fp32.

`rel_err` is max|a-b| / max|b| (error relative to the tensor's dynamic range -- RANGE-relative,
not element-wise), the metric the reference's own "<10^-4" repeatability statement (reference
test.py:16-20) is about; every "1e-4" / "2e-5" / "2e-6" bound in tests/ is in that metric unless it
says otherwise.  `elementwise_rel_percentiles` is the element-wise companion used on depth maps.  Discrete outputs (masks, argmax planes) are compared exactly, with a
tiny allowance for ties that fp32 reassociation can flip (SURVEY.md §7 hard parts)."""
import numpy as np

TOL = 1e-4


def to_np(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
    except ImportError:
        pass
    return np.asarray(x)


def rel_err(a, b):
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)


def assert_close(a, b, tol=TOL, what=""):
    e = rel_err(a, b)
    assert np.isfinite(to_np(a)).all(), f"{what}: non-finite values"
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


def elementwise_rel_percentiles(a, b, floor=1e-6):
    """Element-wise |a-b| / max(|b|, floor) at the 50th / 99th / 99.9th percentile and its maximum -- stricter than
    `rel_err` (which is relative to the tensor's dynamic range, the metric of the reference's own "<1e-4" statement,
    test.py:16-20): used for positive quantities such as depth_pred, where every element has a meaningful scale."""
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    e = (np.abs(a - b) / np.maximum(np.abs(b), floor)).ravel()
    return {"p50": float(np.percentile(e, 50)), "p99": float(np.percentile(e, 99)),
            "p99.9": float(np.percentile(e, 99.9)), "max": float(e.max())}


def mismatch_fraction(a, b):
    a, b = to_np(a), to_np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a != b).mean())


def assert_lowest_cost(lowest, cost_volume, planes_bd, ref_lowest, what=""):
    """lowest_cost = planes[argmax_d cost] (reference cost_volume.py:374-378).  Must be
    self-consistent with OUR volume exactly; vs the reference, argmax may flip only where
    the two top costs tie to within the parity tolerance."""
    lowest, cv, ref_lowest = to_np(lowest), to_np(cost_volume), to_np(ref_lowest)
    planes = to_np(planes_bd)
    idx = cv.argmax(1)
    if planes.ndim == 2:
        own = np.take_along_axis(planes[:, :, None, None] * np.ones_like(cv), idx[:, None], 1)[:, 0]
    else:
        own = np.take_along_axis(planes, idx[:, None], 1)[:, 0]
    assert np.array_equal(lowest, own.astype(lowest.dtype)), f"{what}: lowest_cost != planes[argmax(own volume)]"
    # compare plane INDICES (plane values computed on another device may differ in the last ulp)
    if planes.ndim == 2:
        ref_idx = np.abs(planes[:, :, None, None] - ref_lowest[:, None]).argmin(1)
    else:
        ref_idx = np.abs(planes - ref_lowest[:, None]).argmin(1)
    bad = idx != ref_idx
    if bad.any():
        srt = np.sort(cv, axis=1)
        gap = (srt[:, -1] - srt[:, -2]) / max(np.abs(cv).max(), 1e-30)
        assert (gap[bad] <= 2 * TOL).all(), f"{what}: argmax differs from reference beyond a tie ({bad.sum()} px)"
    return float(bad.mean())


def capture_cv_encoder_levels(store, key="levels"):
    """Forward hook for CVEncoder that leaves its outputs (all levels, in order) in store[key].  Since r06 DepthModel calls the
    module with defer_last=True: the output is (levels 0 .. n-2, finish) and `finish` -- run later, inside the decoder -- yields
    the deepest level; the hook wraps it so that its result is recorded too."""
    def hook(_module, _args, out):
        if isinstance(out, tuple) and len(out) == 2 and (out[1] is None or callable(out[1])):
            levels, finish = out
            store[key] = list(levels)
            if finish is None:
                return None

            def recording_finish():
                last = finish()
                store[key].append(last)
                return last
            return (levels, recording_finish)
        store[key] = list(out)
        return None
    return hook
