import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# (five streams in flight in some tests: a hardware queue each -- read by the HIP runtime when it initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run by the driver with -m gpu)"
    )


_cache = {}


def golden(name):
    """Load tests/golden/<name>.npz (cached)."""
    if name not in _cache:
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            _cache[name] = {k: z[k] for k in z.files}
    return _cache[name]


@pytest.fixture(scope="session")
def load_golden():
    return golden
