import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# RF_PATH_AUTO sends images up to 1024^2 to the line-parallel untiled kernels (launch-bound regime).  Most parity tests
# use small shapes on purpose and mean the TILED kernels when they ask for the automatic path, so the suite switches that
# choice off; the tests of the small-image path ask for it explicitly (path=1) or re-enable it (monkeypatch.delenv).
os.environ.setdefault("RF_SMALL_LIMIT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
