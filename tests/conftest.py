import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _bounded_oracle_threads():
    """The CPU oracle (torch fp32) is the checker in these tests.  On a many-core host torch starts one thread per
    hardware thread, and the small per-read oracle calls of the parity tests then spend their time in thread
    hand-offs (bench.py's probe: 16 threads 30 reads/s, 256 threads 0.5): cap it at the GPU box's CPU share."""
    import torch

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(GOLDEN, "reference_kats.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if needed) and load the C-ABI library.  GPU tests must never fall back silently."""
    from deepchopper_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()
