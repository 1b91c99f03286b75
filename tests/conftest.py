import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_initialises_the_gpu_first(request):
    """A GPU test process holds two HIP runtimes (the one bundled with torch and the one libspmm_hip.so links);
    torch finds no GPU when the other one came up first, so GPU runs bring torch's up before any test."""
    if os.path.exists("/dev/kfd") and any(i.get_closest_marker("gpu") for i in request.session.items):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    yield
