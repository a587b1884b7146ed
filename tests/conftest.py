import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the library's development switches (KLNMF_QTILE=8, KLNMF_COL8, KLNMF_Q8_MONITOR, ...: csrc/ctx.hip.h, DevSwitches) are
# honoured only under KLNMF_DEV=1; the tests use them to reach every kernel variant
os.environ.setdefault('KLNMF_DEV', '1')


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should report skips, not crash.
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
