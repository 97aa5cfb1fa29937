import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_params(tag="dns3"):
    return np.fromfile(os.path.join(GOLDEN, f"params_{tag}.f32"), dtype=np.float32)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, b):
    """max |a-b| / max |b|: the parity metric (north_star: <= 1e-4 for fp32)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="session")
def params_dns3():
    return load_params("dns3")


@pytest.fixture(scope="session")
def params_rand():
    return load_params("rand")
