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


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b)) / den)


def rel_err_rows(a, b, floor=1e-6):
    """Worst row of max|a - b| per row over that row's own max|b|: for fp32 assertions over many latents, where normalising by the
    global maximum would hide a wrong small-magnitude latent (a 100 % error on a series 1e-4 the size of the largest passes 1e-4).
    `floor` (relative to the global scale) keeps an all-zero row from dividing by zero."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    den = np.maximum(np.max(np.abs(b), axis=1), floor * max(float(np.max(np.abs(b))), 1e-300))
    return float(np.max(np.max(np.abs(a - b), axis=1) / den))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def hip_built():
    """Build the HIP library once per session (cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    from multioutputihgp_amd import library_path
    assert os.path.exists(library_path())
    return library_path()
