import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


import avi_talking_amd  # noqa: E402

avi_talking_amd.request_hw_queues(8)     # before any test touches the GPU (explicit: importing the package changes nothing)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import avi_talking_amd.lib as L
    L.load()   # fails loudly if the HIP library has not been built
    return torch.device("cuda:0")
