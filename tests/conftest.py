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


@pytest.fixture(scope="session")
def golden():
    def load(name):
        path = os.path.join(GOLDEN, name + ".npz")
        if not os.path.exists(path):
            pytest.skip(f"golden fixture {name}.npz not generated (tests/golden/make_golden.py)")
        return np.load(path, allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as o
    o.build()
    return o


# SVMC_RECORD_TOLERANCES=<file>: every np.testing.assert_allclose of the run also appends what it OBSERVED -- the largest
# |actual - desired| / (atol + rtol |desired|) (1 = at the tolerance) and the largest plain relative deviation -- with the
# calling test line, so that tolerances can be set from evidence (profiles/r03_observed_tolerances.txt)
if os.environ.get("SVMC_RECORD_TOLERANCES"):
    import inspect

    _orig_allclose = np.testing.assert_allclose

    def _recording_allclose(actual, desired, rtol=1e-7, atol=0, *args, **kwargs):
        try:
            a, d = np.asarray(actual), np.asarray(desired)
            if not (np.iscomplexobj(a) or np.iscomplexobj(d)):
                a, d = a.astype(np.float64), d.astype(np.float64)
            a, d = np.broadcast_arrays(a, d)
            ok = np.isfinite(a) & np.isfinite(d)
            if ok.any():
                diff = np.abs(a[ok] - d[ok])
                used = float(np.max(diff / np.maximum(atol + rtol * np.abs(d[ok]), 1e-300)))
                rel = float(np.max(diff / np.maximum(np.abs(d[ok]), 1e-300)))
                frame = next(f for f in inspect.stack()[1:] if os.sep + "tests" + os.sep in f.filename)
                with open(os.environ["SVMC_RECORD_TOLERANCES"], "a") as fh:
                    fh.write(f"{os.path.basename(frame.filename)}:{frame.lineno} {frame.function} rtol={rtol:g} atol={atol:g} "
                             f"used={used:.3e} max_rel={rel:.3e} n={int(ok.sum())}\n")
        except Exception:                                    # recording must never change a test's outcome
            pass
        return _orig_allclose(actual, desired, rtol, atol, *args, **kwargs)

    np.testing.assert_allclose = _recording_allclose
