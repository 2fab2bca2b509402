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
