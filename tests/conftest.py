import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # developer aid (never set by the driver): CTL_TEST_LIB=<name|path> runs the suite against an A/B build of the kernels
    # (tools/build_variant.sh), e.g. a -DCTL_TUNING build with CTL_X3_PC_MIN_STEPS=1 to force the producer / consumer kernels onto the
    # small test shapes.  The package itself reads no environment variable.
    lib = os.environ.get("CTL_TEST_LIB")
    if lib:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import _variant
        _variant.use_variant(lib)


@pytest.fixture(scope="session")
def golden_cases():
    import torch
    return torch.load(os.path.join(GOLDEN, "cases.pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden_sd():
    import torch
    return torch.load(os.path.join(GOLDEN, "state_dicts_seed0.pt"), weights_only=False)
