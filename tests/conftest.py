import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# The library reads no environment variable.  RF_PATH_AUTO sends images up to 1024^2 to the line-parallel untiled kernels
# (launch-bound regime); most parity tests use small shapes on purpose and mean the TILED kernels when they ask for the
# automatic path, so plans the suite creates WITHOUT explicit flags get rf_filter_desc.flags = RF_PLAN_TILED_ONLY.  The
# shipped default (flags = 0) is what tests/test_shipped_defaults.py runs the same small cases on, and what the tests
# that take the `shipped_defaults` fixture see.
import recfilter_amd.plan as _rf_plan
from recfilter_amd import capi as _rf_capi

_rf_plan.DEFAULT_FLAGS = _rf_capi.RF_PLAN_TILED_ONLY


@pytest.fixture
def plan_flags():
    """plan_flags(f): plans created without explicit flags get rf_filter_desc.flags = f for the rest of the test."""
    saved = _rf_plan.DEFAULT_FLAGS

    def set_flags(flags):
        _rf_plan.DEFAULT_FLAGS = int(flags)
    yield set_flags
    _rf_plan.DEFAULT_FLAGS = saved


@pytest.fixture
def shipped_defaults(plan_flags):
    """The test runs on the defaults a user gets: rf_filter_desc.flags = 0."""
    plan_flags(0)


@pytest.fixture(autouse=True)
def _deterministic_inputs(request):
    """Every test starts from the same generator state (numpy's legacy generator, torch on the host and on the GPU), derived
    from the test's id: a test that draws an input without a seed of its own still sees the same input on every box and in
    every order (one unseeded torch.rand turned a driver run red).  The tests seed their inputs explicitly; this
    is the net under them."""
    import zlib
    import numpy as np
    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    np.random.seed(seed)
    try:
        import torch
        torch.manual_seed(seed)            # seeds the CUDA generators too when a GPU is there
    except Exception:
        pass
    yield


import parity_record as _parity_record

_parity_record.install()      # RF_RECORD_PARITY=<file>: write every metric evaluation down for tests/metric_margin.py


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
