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


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first():
    """A few GPU tests put frames on the device through torch.  torch initialises HIP lazily, and on the GPU boxes that initialisation
    fails ("No HIP GPUs are available") when it comes AFTER this library has been driven through its error-path tests in the same
    process - seen when tests/test_gpu_device_contours.py runs on its own; in the full suite an earlier file makes torch initialise
    first.  So: first thing, when there is a GPU at all."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
