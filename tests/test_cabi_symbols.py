"""
The C-ABI library builds for gfx950, loads on a machine without a GPU, and exports exactly the entry points
include/svmc.h declares (no compute calls here).  Also: the product has no CPU fallback -- asking for the
Monte Carlo path without a GPU fails loudly.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "svmc.h")).read()
    return sorted(set(re.findall(r"SVMC_API\s+(?:const\s+char\s*\*|int)\s*(svmc_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from stochvolmodels_amd import build
    lib = build.build()
    assert os.path.exists(lib)
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (svmc_[a-z0-9_]+)", exported)))
    declared = _declared()
    assert len(declared) >= 30
    assert exported == declared
    L = C.CDLL(lib)
    for name in declared:
        assert hasattr(L, name)
    L.svmc_version.restype = C.c_int
    assert L.svmc_version() == 100


def test_python_binding_covers_the_header():
    from stochvolmodels_amd import _lib
    L = _lib.load()
    assert sorted(L._svmc_symbols) == _declared()


def test_gfx950_code_object_present():
    from stochvolmodels_amd import build
    out = subprocess.run(["strings", "-a", build.build()], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_host_only_entry_points_work_without_gpu():
    from stochvolmodels_amd.engine import payoff_finalize
    sums = np.array([3.0, 5.0, 4.0, 0.0, 0.0, 0.0])          # strike 0: sum d = 3, sum d^2 = 5, n = 4
    prices, stderrs = payoff_finalize(sums, np.array([1.0, 0.0]), 0.5, 16.0)
    np.testing.assert_allclose(prices[0], 0.5 * (1.0 + 0.75))
    np.testing.assert_allclose(stderrs[0], 0.5 * np.sqrt(5 / 4 - 0.75 ** 2) / 4.0)
    assert np.isnan(prices[1])                                 # 0/0 like nanmean of an all-NaN slice


def test_random_stream_version_is_one_number_everywhere():
    """the stream version the library reports, the header's define, the Python constant and the table both sides were generated
    from (version 4: the lattice point of a word is the signed integer itself, SVMC_ICDF_HALF_LATTICE 0) agree"""
    import stochvolmodels_amd as sv
    from stochvolmodels_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "svmc.h")).read()
    declared = int(re.search(r"#define SVMC_RNG_STREAM_VERSION (\d+)", header).group(1))
    assert declared == sv.RNG_STREAM_VERSION == _lib.load().svmc_rng_stream_version() == 4
    for rel in (("stochvolmodels_amd", "csrc", "svmc_icdf_table.h"), ("oracle", "svo_icdf_table.h")):
        table = open(os.path.join(root, *rel)).read()
        assert "#define SVMC_ICDF_HALF_LATTICE 0" in table and "#define SVMC_ICDF_RAW 1" in table
        assert f"version {declared}" in table.splitlines()[1]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback():
    import stochvolmodels_amd as sv
    from stochvolmodels_amd._lib import SvmcError
    with pytest.raises(SvmcError):
        sv.LogSVPricer().simulate_terminal_values(sv.LOGSV_BTC_PARAMS, nb_path=8)
    with pytest.raises(SvmcError):
        sv.compute_mc_vars_payoff(x0=np.zeros(4), sigma0=np.ones(4), qvar0=np.zeros(4), ttm=1.0, forward=1.0,
                                  strikes_ttm=np.array([1.0]), optiontypes_ttm=np.array(["C"]))


@pytest.mark.parametrize("name", ["price_chain", "price_chain_rccl", "price_chain_multi", "calibration_objective"])
def test_c_example_compiles_and_links(tmp_path, name):
    """the examples are plain-C hosts of the library -- one GPU, a process per GPU over RCCL, several GPUs from one process, a calibration's objective: each
    must compile with gcc -Wall -Werror against include/svmc.h and link against libsvmc.so (they are RUN by the gpu suite)"""
    from stochvolmodels_amd import build
    lib = build.build()
    exe = str(tmp_path / name)
    libdir = os.path.dirname(lib)
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L" + libdir, "-lsvmc",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"], check=True)
    assert os.path.exists(exe)
