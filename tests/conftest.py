import os
import sys

import pytest

# PyTorch ships its own copy of the HIP runtime.  On this image it must be the first HIP runtime to initialise in a
# process: if libmalva_hip.so (linked against /opt/rocm's runtime) touches the GPU first, torch later reports
# "No HIP GPUs are available".  The tests that use both (stream sharing, aliased counters) rely on this import order;
# bench.py imports torch first for the same reason.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
