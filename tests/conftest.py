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


TOL = 1e-4    # the contract (north_star: 1e-4 relative fp32)
REG = 2e-5    # regression guard on whole-forward outputs: ~5x what the fp32-class path measures (2.5e-6 .. 4e-6 against
              # the oracle, round 3).  The contract alone has 40x slack: a dense 3x3 that lost its hi.lo / lo.hi products
              # (error ~2^-16 per product) would still pass it.  tests/test_gpu_precision.py holds the path against float64.
REG_STAGE = 1e-5   # ... on single stage boundaries (en0 .. de4)


def check_parity(a, b, what="", reg=REG):
    """Contract assert (1e-4) AND regression assert (reg) on max|a-b| / max|b|; returns the error."""
    e = rel_err(a, b)
    assert e < TOL, f"{what}: {e:.3e} breaks the 1e-4 contract"
    assert e < reg, f"{what}: {e:.3e} is inside the contract but above the regression guard {reg:g}"
    return e


@pytest.fixture(scope="session")
def params_dns3():
    return load_params("dns3")


@pytest.fixture(scope="session")
def params_rand():
    return load_params("rand")


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """The checkers (oracle/torch_port.py) run ATen on the host: with the 128+ threads of a GPU box's CPU the
    tiny convolutions of this model crawl (minutes instead of seconds), so the CPU side is capped."""
    try:
        import torch
        torch.set_num_threads(min(8, torch.get_num_threads()))
    except Exception:
        pass
    yield
