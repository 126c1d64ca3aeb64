import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def sr_option():
    """`sr_option("SR_PW_NT", 2)`: sets a switch of the library's option table (include/simplerecon_hip.h, SR_OPT_*) for the
    rest of the test and restores every switch it touched afterwards.  (r01-r04 tests used monkeypatch.setenv: the library
    no longer reads the environment after its first option access.)"""
    from simplerecon_amd import _lib
    saved = []

    def set_(name, value):
        saved.append((name, _lib.set_option(name, value)))
    yield set_
    for name, prev in reversed(saved):
        _lib.set_option(name, prev)
