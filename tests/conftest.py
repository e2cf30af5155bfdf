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


def has_gpu():
    try:
        import bohip
        from bohip import _lib

        return _lib.load().bohip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def orc():
    from oracle.oracle import COracle

    return COracle()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def synth(N, d, R, seed=0):
    """BASELINE.md synthetic recipe."""
    rng = np.random.default_rng(seed)
    X = rng.random((N, d))
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    Xs = rng.random((R, d))
    return X, y, Xs


def var_tol(ref, N, s2f, rel=1e-6):
    """|d sigma^2| <= rel*|ref| + c*N*eps*s_f^2 : sigma^2 = s_f^2 - v'v cancels catastrophically near
    observations, so a pure relative bound is unattainable by ANY float64 implementation."""
    return rel * np.abs(ref) + 64 * N * np.finfo(np.float64).eps * s2f
