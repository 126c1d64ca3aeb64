"""End-to-end data flow: keyframe selection -> DepthModel.forward -> TSDF fusion (examples/stream_fusion.py)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_stream_fusion_example_runs():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "stream_fusion.py")
    spec = importlib.util.spec_from_file_location("stream_fusion", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    predicted, touched = mod.run(frames=90, height=96, width=128, verbose=False)
    assert predicted >= 3, "the synthetic path should produce full 8-view tuples"
    assert touched > 0, "random-weight depths still fall inside the fusion volume"
