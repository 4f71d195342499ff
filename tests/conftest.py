import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_cases():
    import torch
    return torch.load(os.path.join(GOLDEN, "cases.pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden_sd():
    import torch
    return torch.load(os.path.join(GOLDEN, "state_dicts_seed0.pt"), weights_only=False)
