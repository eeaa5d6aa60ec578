"""pytest configuration.

Markers
  gpu   needs a real MI355X (driver runs `-m gpu` on the GPU box, `-m "not gpu"` in the CPU container)
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a gpu-marked test on a box without a GPU is an ERROR in the selection, not a silent pass:
    # skip only when the user did not ask for gpu tests explicitly.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return  # asked for gpu tests without a gpu: let them fail loudly
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_everything():
    """Every test starts from the same RNG state (tests that need a specific seed set their own): no flaky tolerances."""
    np.random.seed(20260928)
    try:
        import torch
        torch.manual_seed(20260928)
    except Exception:
        pass
    yield


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def formula_tensor(shape, freq, phase):
    """Same closed form as tests/golden/make_golden.py (inputs that are not stored in the fixtures)."""
    i = np.arange(int(np.prod(shape)), dtype=np.float64)
    return (np.sin(freq * i + phase) + 0.5 * np.cos(0.013 * i)).astype(np.float32).reshape(shape)


def csr_from(d, key, n=None, dtype=np.float64):
    import scipy.sparse as sp
    indptr = d[key + "_indptr"]
    n = len(indptr) - 1 if n is None else n
    return sp.csr_matrix((d[key + "_data"].astype(dtype), d[key + "_indices"], indptr), shape=(n, n))


def close_scaled(got, ref, rtol=1e-5, atol_scale=2e-6):
    """SURVEY.md §8c fp32 SpMM tolerance: |got-ref| <= rtol*|ref| + atol_scale*max|ref|."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    tol = rtol * np.abs(ref) + atol_scale * max(1.0, float(np.abs(ref).max(initial=0.0)))
    bad = np.abs(got - ref) > tol
    assert not bad.any(), "max abs err %.3e (tol at worst %.3e), %d/%d out of tolerance" % (
        np.abs(got - ref).max(), tol[bad].min() if bad.any() else 0, bad.sum(), bad.size)
