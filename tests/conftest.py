import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "point-cloud-processing_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a device must fail loudly, not skip silently; on a
    # CPU-only host running the whole suite (no -m), GPU tests are skipped.
    if _have_gpu():
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
