"""
Accuracy of the hand-written fp64 functions of stochvolmodels_amd/csrc/svmc_math.h, host build (g++), against
80-bit libm.  The host build emulates the v_rcp_f64 / v_rsq_f64 seeds with single-precision reciprocals (2^-24,
what the hardware delivers: tools/ubench/math_probe.hip), so the refinement steps see their worst case.
Stated bounds: exp, sin, cos <= 2 ULP (measured 1.1 / 1.6 / 1.0); -log <= 3 ULP (measured 2.8: its quotient
uses a single Newton step, which is what bounds it); sqrt and 1/x correctly rounded on the sampled ranges.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("probe") / "libmath_probe.so")
    flags = ["-mfma"] if "fma" in open("/proc/cpuinfo").read() else []
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", *flags,
                    "-I" + os.path.join(ROOT, "stochvolmodels_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "math_probe.cpp"), "-o", so], check=True)
    return C.CDLL(so)


def _call(lib, fn, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    getattr(lib, fn)(x.ctypes.data_as(DP), y.ctypes.data_as(DP), C.c_size_t(x.size))
    return y


def _ulp(y, ref):
    ref = np.asarray(ref, dtype=np.longdouble)
    return float(np.max(np.abs((y.astype(np.longdouble) - ref) / np.spacing(np.abs(ref.astype(np.float64))))))


N = 400_000


def test_exp(probe):
    rng = np.random.default_rng(0)
    for lo, hi in ((-1, 1), (-30, 30), (-700, 700)):
        x = rng.uniform(lo, hi, N)
        assert _ulp(_call(probe, "probe_exp", x), np.exp(x.astype(np.longdouble))) <= 2.0
    assert _call(probe, "probe_exp", np.array([800.0]))[0] == np.inf          # v_ldexp saturation semantics
    assert _call(probe, "probe_exp", np.array([-800.0]))[0] == 0.0
    assert np.isnan(_call(probe, "probe_exp", np.array([np.nan]))[0])


def test_exp_full(probe):
    """exp_full: the exponential of the payoff reductions -- any double in, the libm answer out: <= 2 ULP on the finite
    range incl. the denormal results below e^-708, inf above the overflow point, +0 below the last denormal, +-inf and
    NaN handled (terminal log-returns of overflowed paths reach it)"""
    rng = np.random.default_rng(3)
    for lo, hi in ((-1, 1), (-30, 30), (-708, 709.7)):
        x = rng.uniform(lo, hi, N)
        assert _ulp(_call(probe, "probe_exp_full", x), np.exp(x.astype(np.longdouble))) <= 2.0
    x = rng.uniform(-745.0, -708.0, 20000)                              # gradual underflow: denormal results
    got, ref = _call(probe, "probe_exp_full", x), np.exp(x)
    assert np.all(np.abs(got - ref) <= 2.0 * 4.9406564584124654e-324 + 4e-16 * ref)
    edge = np.array([709.78, 709.79, 745.9, 746.0, 746.1, 1e4, 1e300, np.inf, -745.13, -745.2, -746.0, -746.1, -1e4, -1e300, -np.inf, 0.0])
    with np.errstate(over="ignore", under="ignore"):
        want = np.exp(edge)
    got = _call(probe, "probe_exp_full", edge)
    for g_, w_, x_ in zip(got, want, edge):
        assert (g_ == w_) or (np.isfinite(w_) and w_ > 0 and abs(g_ / w_ - 1) < 1e-15) or (w_ < 1e-320 and abs(g_ - w_) < 1e-322), (x_, g_, w_)
    assert np.isnan(_call(probe, "probe_exp_full", np.array([np.nan]))[0])


def test_exp_table(probe):
    """exp_tab (256-entry table, quadratic tail, one-constant reduction: the exponential of the issue-bound stepping
    kernels, whose arguments are log-volatilities).  <= 1.5 ULP on |x| <= 1, <= 3 ULP on |x| <= 5; the one-constant
    reduction adds 3.4e-17 |x| relative beyond that (2.4e-14 at the ends of the double range)."""
    rng = np.random.default_rng(10)
    for lo, hi, bound in ((-1, 1, 1.5), (-5, 5, 3.0), (-30, 30, 11.0)):
        x = rng.uniform(lo, hi, N)
        assert _ulp(_call(probe, "probe_exp_tab", x), np.exp(x.astype(np.longdouble))) <= bound
    x = rng.uniform(-700, 700, N)
    rel = np.abs(_call(probe, "probe_exp_tab", x).astype(np.longdouble) / np.exp(x.astype(np.longdouble)) - 1)
    assert float(rel.max()) <= 3e-14
    x = np.arange(-400, 401) * (np.log(2.0) / 256)                  # the table nodes, both sides of every rounding tie
    for eps in (0.0, 1e-17, -1e-17, 1.35e-3, -1.35e-3):
        assert _ulp(_call(probe, "probe_exp_tab", x + eps), np.exp((x + eps).astype(np.longdouble))) <= 1.5
    assert _call(probe, "probe_exp_tab", np.array([800.0]))[0] == np.inf
    assert _call(probe, "probe_exp_tab", np.array([-800.0]))[0] == 0.0
    assert _call(probe, "probe_exp_tab", np.array([0.0]))[0] == 1.0
    assert np.isnan(_call(probe, "probe_exp_tab", np.array([np.nan]))[0])


def test_exp2u_table(probe):
    """exp2u_tab(y) = 2^(y/256), the exponential of the LogSV stepping kernels (log-volatility carried in units of
    ln2/256, reduction exact): <= 1.1 ULP whatever the size of the argument"""
    rng = np.random.default_rng(12)
    two = np.longdouble(2.0)
    for lo, hi in ((-400, 400), (-3000, 3000), (-250000, 250000)):
        y = rng.uniform(lo, hi, N)
        assert _ulp(_call(probe, "probe_exp2u_tab", y), two ** (y.astype(np.longdouble) / 256)) <= 1.1
    y = np.arange(-2000, 2001).astype(np.float64)                    # the table nodes and the rounding ties between them
    for eps in (0.0, 0.5, -0.5, 0.4999999, -0.4999999):
        assert _ulp(_call(probe, "probe_exp2u_tab", y + eps), two ** ((y + eps).astype(np.longdouble) / 256)) <= 1.1
    assert _call(probe, "probe_exp2u_tab", np.array([0.0]))[0] == 1.0
    assert _call(probe, "probe_exp2u_tab", np.array([256.0 * 1100]))[0] == np.inf          # v_ldexp saturation semantics
    assert _call(probe, "probe_exp2u_tab", np.array([-256.0 * 1200]))[0] == 0.0
    assert np.isnan(_call(probe, "probe_exp2u_tab", np.array([np.nan]))[0])
    assert np.isnan(_call(probe, "probe_exp2u_tab", np.array([np.inf]))[0])                # as exp_tab: inf - inf in the reduction


def test_neg_log(probe):
    rng = np.random.default_rng(1)
    u = rng.integers(0, 2 ** 52, N).astype(np.float64) * 2.0 ** -52 + 2.0 ** -53       # the RNG lattice
    assert _ulp(_call(probe, "probe_neg_log", u), -np.log(u.astype(np.longdouble))) <= 3.0
    u = np.concatenate([2.0 ** -rng.uniform(0, 53, N), 1 - 2.0 ** -rng.uniform(1, 53, N),
                        [2.0 ** -53, 1 - 2.0 ** -53, 0.5, np.sqrt(0.5)]])
    u = u[(u > 0) & (u < 1)]
    assert _ulp(_call(probe, "probe_neg_log", u), -np.log(u.astype(np.longdouble))) <= 3.0


def test_neg_log_table(probe):
    """the table-assisted -ln of the RNG, called on u in (0,1) only: <= 2 ULP (1.1 typical; the intervals next to
    the c = 1 one cancel log_c against log1p(f) and reach 1.8) on the RNG lattice, in both tails (u -> 0, and
    u -> 1 where the c = 1 interval keeps the RELATIVE accuracy) and at both ends of every table interval below 1.
    (Above 1 the doubled interval width lets the cancellation reach ~9 ULP of a ~1e-4 result, 1e-19 absolute;
    Heston QE, the only caller with such arguments, uses the divide-based neg_log.)"""
    rng = np.random.default_rng(4)
    u = rng.integers(0, 2 ** 52, N).astype(np.float64) * 2.0 ** -52 + 2.0 ** -53
    assert _ulp(_call(probe, "probe_neg_log_tab", u), -np.log(u.astype(np.longdouble))) <= 2.0
    u = np.concatenate([2.0 ** -rng.uniform(0, 53, N), 1 - 2.0 ** -rng.uniform(1, 53, N),
                        [2.0 ** -53, 1 - 2.0 ** -53, 0.5, np.sqrt(0.5)]])
    u = u[(u > 0) & (u < 1)]
    assert _ulp(_call(probe, "probe_neg_log_tab", u), -np.log(u.astype(np.longdouble))) <= 2.0
    hi = (0x3FE6A09E + np.arange(512, dtype=np.uint64) * 2048)
    ends = np.concatenate([(hi << np.uint64(32)), ((hi + np.uint64(2047)) << np.uint64(32)) | np.uint64(0xFFFFFFFF)]).view(np.float64)
    for scale in (1.0, 0.5, 2.0 ** -20):
        e = ends * scale
        e = e[e < 1.0]
        assert _ulp(_call(probe, "probe_neg_log_tab", e), -np.log(e.astype(np.longdouble))) <= 2.0
    assert _call(probe, "probe_neg_log_tab", np.array([1.0]))[0] == 0.0
    v = 2.0 ** rng.uniform(-30, 30, N)                              # general arguments: absolute accuracy only
    err = np.abs(_call(probe, "probe_neg_log_tab", v).astype(np.longdouble) + np.log(v.astype(np.longdouble)))
    assert float(err.max()) <= 4e-15


def test_sqrt_and_rcp(probe):
    rng = np.random.default_rng(2)
    t = 2.0 ** rng.uniform(-60, 9, N)
    assert _ulp(_call(probe, "probe_sqrt", t), np.sqrt(t.astype(np.longdouble))) <= 1.0
    a = 2.0 ** rng.uniform(-20, 20, N)
    assert _ulp(_call(probe, "probe_rcp", a), 1 / a.astype(np.longdouble)) <= 1.0
    # the radius of the Box-Muller pair: the Goldschmidt step alone.  1.5 e^2 of the seed error e: 2^-47 on the
    # hardware seed (2^-24.2, tools/ubench/math_probe.hip), 2^-45 with the float seed of this host build (2^-23)
    t = np.concatenate([2.0 ** rng.uniform(-53, 5.3, N), [1.1102230246251565e-16, 36.7368005696771]])
    rel = np.abs(_call(probe, "probe_sqrt_1g", t).astype(np.longdouble) / np.sqrt(t.astype(np.longdouble)) - 1)
    assert float(rel.max()) <= 2.0 ** -45


def test_normal_icdf32(probe):
    """normal_icdf32: one N(0,1) variate from one raw 32-bit word (random stream version 4) -- the product's own source
    compiled for the host must be the CPU twin's function bit for bit (oracle svo_normal_from_word restates it in C), and
    within the table's stated error of Phi^-1"""
    from scipy.special import ndtri
    from oracle import oracle
    rng = np.random.default_rng(13)
    U32 = C.POINTER(C.c_uint32)
    w = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    edges = np.concatenate([np.arange(-3, 4, dtype=np.int64) + (1 << e) for e in range(31)])
    w = np.concatenate([w, edges.astype(np.uint32), (-edges).astype(np.uint32), [0, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000]]).astype(np.uint32)
    z = np.empty(w.size)
    probe.probe_normal_icdf32(w.ctypes.data_as(U32), z.ctypes.data_as(DP), C.c_size_t(w.size))
    twin = np.array([oracle.normal_from_word(int(v)) for v in w])
    np.testing.assert_array_equal(z, twin)
    t = w.view(np.int32).astype(np.float64)                # stream version 4: the lattice point is the signed word itself
    with np.errstate(divide="ignore"):
        exact = np.where(t == 0.0, 0.0, np.copysign(-ndtri(np.abs(t) * 2.0 ** -32), t))
    assert float(np.max(np.abs(z - exact))) <= 1e-9


def test_log_state(probe):
    """log_state: ln of a state variable with its constants in scalar registers -- <= 4 ULP on normal arguments (3.4 measured
    where k ln2 and ln m partly cancel), and the
    special values the generators can meet at a slice start: denormals, 0 -> -inf, +inf -> +inf, NaN and negatives -> NaN"""
    rng = np.random.default_rng(14)
    for lo, hi in ((-3, 3), (-60, 60), (-1000, 1000)):
        s = 2.0 ** rng.uniform(lo, hi, N // 4)
        assert _ulp(_call(probe, "probe_log_state", s), np.log(s.astype(np.longdouble))) <= 4.0
    tiny = 2.0 ** rng.uniform(-1074, -1000, 2000)
    assert _ulp(_call(probe, "probe_log_state", tiny), np.log(tiny.astype(np.longdouble))) <= 3.0
    got = _call(probe, "probe_log_state", np.array([0.0, np.inf, np.nan, -1.0, -np.inf, 1.0, 5e-324]))
    assert got[0] == -np.inf and got[1] == np.inf and np.isnan(got[2]) and np.isnan(got[3]) and np.isnan(got[4])
    assert got[5] == 0.0 and abs(got[6] - np.log(5e-324)) < 1e-12


def test_ode_rows_equal_the_twins_right_hand_side(tmp_path_factory, oracle):
    """svmc_ode.h: the LogSV coefficient ODE one component per lane (make_ode_lane + ode_rhs_lane, host build) against the CPU
    twin's right-hand side (oracle/svmc_oracle_analytic.c ode_rhs): both measures, both expansion orders, backbones"""
    so = str(tmp_path_factory.mktemp("probe") / "libode_probe.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                    "-I" + os.path.join(ROOT, "stochvolmodels_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "ode_probe.cpp"), "-o", so], check=True)
    lib = C.CDLL(so)
    lib.probe_ode_rhs_lanes.argtypes = [C.c_double] * 5 + [C.c_int, C.c_int, C.c_double] + [DP] * 4
    rng = np.random.default_rng(11)
    worst = 0.0
    for trial in range(400):
        theta, k1, k2 = rng.uniform(0.2, 1.5), rng.uniform(0.5, 5.0), rng.uniform(0.0, 12.0)
        beta, volvol, eta = rng.uniform(-1.0, 0.5), rng.uniform(0.3, 2.5), rng.uniform(0.8, 1.2)
        spot, order = bool(trial & 1), 1 + ((trial >> 1) & 1)
        phi = complex(rng.choice([-0.5, 0.5, rng.uniform(-1, 1)]), rng.uniform(-40, 40))
        psi = complex(rng.uniform(-1, 1), rng.uniform(-40, 40)) if trial % 3 == 0 else 0j
        A = rng.normal(size=5) * 3.0 + 1j * rng.normal(size=5) * 3.0
        if order == 1:
            A[3:] = 0.0
        want = oracle.logsv_ode_rhs(phi, psi, A, theta, k1, k2, beta, volvol, spot, order, eta)
        got = np.empty(5, dtype=np.complex128)
        ph, ps = np.array([phi.real, phi.imag]), np.array([psi.real, psi.imag])
        a = np.ascontiguousarray(A)
        lib.probe_ode_rhs_lanes(theta, k1, k2, beta, volvol, int(spot), order, eta, ph.ctypes.data_as(DP), ps.ctypes.data_as(DP),
                                a.view(np.float64).ctypes.data_as(DP), got.view(np.float64).ctypes.data_as(DP))
        scale = np.max(np.abs(want)) + 1e-300
        worst = max(worst, float(np.max(np.abs(got - want)) / scale))
    assert worst < 1e-13, worst
