import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "cosmology-model-fit_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory has a hyphen in its name, so it is imported through importlib.
    A fresh checkout has no built library (it is git-ignored): compile it once, as __graft_entry__.build() does."""
    pkg = importlib.import_module(PKG_NAME)
    if not os.path.exists(pkg._lib.LIB_PATH):
        pkg.build()
    return pkg


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def synthetic_cov(sigma, seed=0, rank=40, amp=0.01):
    """Same recipe as tests/golden/generate_golden.py::synthetic_cov (the 20 MB matrix is not stored)."""
    rng = np.random.default_rng(seed)
    A = amp * rng.standard_normal((sigma.size, rank))
    return np.diag(sigma**2) + A @ A.T


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def pantheon_golden():
    """Golden vectors of sn/pantheon.py + the regenerated synthetic covariance and its Cholesky factor."""
    from scipy.linalg import cho_factor

    g = dict(golden("sn_pantheon"))
    g["cov"] = synthetic_cov(g["sigma"])
    # cho_factor leaves garbage above the diagonal, exactly as sn/pantheon.py:14
    g["chol"] = cho_factor(g["cov"], lower=True)[0]
    return g
