import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle (PyTorch-CPU) collapses when oversubscribed: the GPU box's host shows 256 cores, its share is 16
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return torch.device("cuda:0")
