import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    has_gpu = _has_gpu()
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU in this container"))


@pytest.fixture(autouse=True)
def _reload_developer_knobs():
    """The library reads its developer knobs (ICV_FORCE_GENERIC, ICV_NO_SD, ...) once; tests that switch kernels through
    the environment call ``_lib.load().icv_developer_knobs_reload()`` after changing it, and this fixture re-reads the
    (restored) environment after every test so that no knob leaks into the next one."""
    yield
    try:
        from infercnvpy_amd import _lib

        if _lib._lib is not None:
            _lib._lib.icv_developer_knobs_reload()
    except Exception:
        pass
