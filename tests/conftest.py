import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """The loaded libnbp_hip.so; GPU tests fail (not skip) when it is missing."""
    import torch
    from nextbestpath_amd import _lib
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    return _lib.lib()


@pytest.fixture(scope="session")
def nbp_weights():
    from nextbestpath_amd.utility.synthetic import make_nbp_state_dict
    return make_nbp_state_dict(9)


@pytest.fixture(scope="module", autouse=True)
def _thaw_frozen_objects():
    """Rollout setup freezes its long-lived objects out of the cyclic collector (testers/nbp_planning.py::_settle_gc).  A test
    session builds hundreds of rollouts: after every module the frozen set is thawed and collected, so that objects in reference
    cycles (and the device tensors they hold) do not outlive their tests."""
    yield
    import gc
    gc.unfreeze()
    gc.collect()
