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


def seeded_parameters(module, seed):
    """Same helper as tests/golden/make_golden.py: parameters from numpy's PCG64 stream, visited in sorted-name order
    (the width-128 fixtures store expected outputs only)."""
    import torch
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(module.named_parameters())):
            rng = np.random.default_rng(seed * 1000 + k)
            bound = 1.0 / np.sqrt(p.shape[-1]) if p.dim() > 1 else 0.1
            v = rng.uniform(-bound, bound, size=tuple(p.shape)).astype(np.float32)
            if name.endswith("norm.weight"):
                v = v * 5.0 + 1.0
            p.copy_(torch.from_numpy(v).to(p.device))


def check_sampled_tensor(g, key, got, rtol, atol_scale):
    """Compare a tensor with its golden record written by make_golden.put_tensor: the full tensor when stored, else 256
    sampled entries plus the sum / abs-sum checksums.  Tolerance: rtol*|ref| + atol_scale*max|ref| per entry."""
    got = np.asarray(got, dtype=np.float64)
    if key in g.files:
        ref = g[key].astype(np.float64)
        scale = max(1e-30, float(np.abs(ref).max(initial=0.0)))
        err = np.abs(got - ref)
        assert (err <= rtol * np.abs(ref) + atol_scale * scale).all(), (key, float(err.max()), scale)
    else:
        ref = g[key + "__vals"].astype(np.float64)
        vals = got.reshape(-1)[g[key + "__pick"]]
        scale = max(1e-30, float(np.abs(ref).max(initial=0.0)))
        err = np.abs(vals - ref)
        assert (err <= rtol * np.abs(ref) + atol_scale * scale).all(), (key, float(err.max()), scale)
    abssum = float(g[key + "__abssum"])
    # checksums: the sum of N entries each within the tolerance above
    assert abs(np.abs(got).sum() - abssum) <= 10 * rtol * abssum + 1e-12, (key, "abssum", np.abs(got).sum(), abssum)
    assert abs(got.sum() - float(g[key + "__sum"])) <= 10 * rtol * abssum + 1e-12, (key, "sum")


def check_close(got, want, rtol, atol, what=""):
    """assert_allclose that also REPORTS how much of the tolerance was used (run with -s): tolerances in the GPU tests are
    set to about ten times the worst observed use, not to whatever passes."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    used = float((err / (atol + rtol * np.abs(want))).max(initial=0.0))
    print("  [tol] %-46s max |err| %.3e  = %.3f of (rtol %g, atol %g)" % (what, float(err.max(initial=0.0)), used, rtol, atol))
    assert used <= 1.0, "%s: max |err| %.3e is %.2f x the tolerance (rtol %g, atol %g)" % (what, err.max(), used, rtol, atol)
