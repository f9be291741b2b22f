import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds libdp_oracle.so on first use."""
    from oracle import dp_oracle
    dp_oracle.build()
    dp_oracle.lib()
    return dp_oracle


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    return torch


@pytest.fixture()
def ctx():
    # torch first: it brings its own libamdhip64, and the HIP runtime that is loaded first serves the whole
    # process; with the system one loaded first (through librt_dp.so) torch does not see the GPU
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    from raytracer_amd import capi
    c = capi.Context()
    yield c
    c.close()
