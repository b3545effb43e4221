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
