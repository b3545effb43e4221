"""Container-only stand-in used by tests/golden/make_golden.py to import the reference in NumPy mode.

`njit` is the identity decorator, i.e. every reference function runs as its `.py_func`, the route the
reference's own tests assert equal to the compiled route at 1e-14
(reference tests/test_logsv_characterization.py:403-404, tests/test_numerical_utilities.py:110-111).
Never imported by the product, the oracle, or any test.
"""
import contextlib


def njit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


jit = njit
prange = range


@contextlib.contextmanager
def objmode(*a, **k):
    yield


class _T:
    def __getattr__(self, n):
        return self

    def __call__(self, *a, **k):
        return self

    def __getitem__(self, k):
        return self


types = float64 = int64 = complex128 = boolean = _T()
from . import typed  # noqa: E402,F401
