import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and libxflate_amd.so is linked against the
# system's; whichever is loaded first serves both, and torch finds no GPU when it is not its own ("No HIP GPUs are
# available").  Some GPU tests use torch for device buffers, so it goes first whatever subset of the tests runs.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle
