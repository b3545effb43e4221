"""
ctypes front-end of the CPU oracle (oracle/svmc_oracle.c) plus a NumPy restatement of the same path.

TEST INFRASTRUCTURE ONLY.  The product package (stochvolmodels_amd) never imports this module; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker / CPU baseline.

Parity pinning: tests/test_oracle_golden.py checks every function here against the golden vectors
made by tests/golden/make_golden.py from the unmodified Python reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsvmc_oracle.so")

CALL, PUT, INV_CALL, INV_PUT = 0, 1, 2, 3
LOG_RETURN, Q_VAR, SIGMA = 1, 2, 3
HESTON_EULER_FLOOR, HESTON_QE = 0, 1
TYPE_CODES = {"C": CALL, "P": PUT, "IC": INV_CALL, "IP": INV_PUT}

_dp = C.POINTER(C.c_double)
_lib = None


def build(force: bool = False) -> str:
    """compile oracle/libsvmc_oracle.so with the committed Makefile (gcc only)."""
    src_time = max(os.path.getmtime(os.path.join(_HERE, f))
                   for f in ("svmc_oracle.c", "svmc_oracle_analytic.c", "svmc_oracle_rough.c", "svmc_oracle.h", "svo_icdf_table.h",
                             "Makefile"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < src_time:
        subprocess.run(["make", "-C", _HERE, "-B", "libsvmc_oracle.so"], check=True, capture_output=True)
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        sz, i32, u32, u64, f64 = C.c_size_t, C.c_int, C.c_uint32, C.c_uint64, C.c_double
        L.svo_set_time_grid.argtypes = [f64, i32, C.POINTER(i32), _dp]
        L.svo_logsv_terminal_w.argtypes = [sz, i32, f64, _dp, _dp, _dp, f64, f64, f64, f64, f64, f64, i32,
                                           _dp, _dp, sz]
        L.svo_heston_terminal_w.argtypes = [sz, i32, f64, _dp, _dp, _dp, f64, f64, f64, f64, _dp, _dp, sz]
        L.svo_heston_qe_terminal_w.argtypes = [sz, i32, f64, _dp, _dp, _dp, f64, f64, f64, f64,
                                               _dp, _dp, _dp, sz]
        L.svo_payoff.argtypes = [sz, _dp, _dp, f64, f64, f64, sz, _dp, C.POINTER(C.c_int8), i32, _dp, _dp]
        L.svo_payoff.restype = i32
        L.svo_philox4x32_10.argtypes = [C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
        L.svo_fill_normals.argtypes = [u64, u32, u64, u32, sz, i32, _dp, _dp, sz]
        L.svo_normal_from_word.argtypes = [u32]
        L.svo_normal_from_word.restype = f64
        L.svo_fill_normals_stream.argtypes = [u64, u32, u32, u64, u32, sz, i32, _dp, _dp, sz]
        L.svo_fill_uniforms.argtypes = [u64, u32, u64, u32, sz, i32, _dp, sz]
        L.svo_logsv_terminal_rng.argtypes = [sz, i32, f64, _dp, _dp, _dp, f64, f64, f64, f64, f64, f64, i32,
                                             u64, u32, u64, u32]
        L.svo_heston_terminal_rng.argtypes = [sz, i32, f64, _dp, _dp, _dp, f64, f64, f64, f64, i32,
                                              u64, u32, u64, u32]
        L.svo_logsv_vol_paths.argtypes = [_dp, sz, sz, i32, f64, f64, f64, f64, f64, f64, f64, i32, _dp, sz, u64, u32, u64]
        L.svo_logsv_vol_paths.restype = None
        L.svo_rough_logsv_terminal_w.argtypes = [sz, i32, f64, i32, _dp, _dp, _dp, f64, f64, f64, f64, f64, _dp, _dp, _dp,
                                                 _dp, _dp, sz]
        L.svo_rough_logsv_terminal_w.restype = None
        L.svo_logsv_mgf_grid.argtypes = [sz, _dp, _dp, f64, f64, f64, f64, f64, f64, f64, i32, i32, f64, _dp, _dp, f64, f64]
        L.svo_logsv_mgf_grid.restype = None
        L.svo_logsv_ode_rhs.argtypes = [f64, f64, f64, f64, f64, i32, i32, f64, _dp, _dp, _dp, _dp]
        L.svo_logsv_ode_rhs.restype = None
        L.svo_heston_mgf_grid.argtypes = [sz, _dp, _dp, f64, f64, f64, f64, f64, f64, _dp, _dp, i32, _dp]
        L.svo_heston_mgf_grid.restype = None
        L.svo_mgf_qvar_slice.argtypes = [sz, _dp, _dp, f64, sz, _dp, C.POINTER(C.c_int8), f64, _dp]
        L.svo_mgf_qvar_slice.restype = i32
        L.svo_mgf_vanilla_slice.argtypes = [sz, _dp, _dp, f64, sz, _dp, C.POINTER(C.c_int8), f64, i32, _dp]
        L.svo_mgf_vanilla_slice.restype = i32
        for name in ("svo_set_time_grid", "svo_logsv_terminal_w", "svo_heston_terminal_w",
                     "svo_heston_qe_terminal_w", "svo_philox4x32_10", "svo_fill_normals", "svo_fill_normals_stream",
                     "svo_fill_uniforms", "svo_logsv_terminal_rng", "svo_heston_terminal_rng"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def effective_cores() -> int:
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_threads(n: int) -> int:
    lib().svo_set_threads.argtypes = [C.c_int]
    lib().svo_set_threads.restype = C.c_int
    return lib().svo_set_threads(int(n))


def _state(*arrs) -> Tuple[np.ndarray, ...]:
    return tuple(np.array(a, dtype=np.float64, order="C", copy=True) for a in arrs)


def _w(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2
    return a


def type_codes(optiontypes: Sequence[str]) -> np.ndarray:
    """'C','P','IC','IP' -> int8 codes; unknown -> ValueError like utils/mc_payoffs.py:84."""
    out = np.empty(len(optiontypes), dtype=np.int8)
    for i, t in enumerate(optiontypes):
        t = str(t)
        if t not in TYPE_CODES:
            raise ValueError("unknown option payoff code")
        out[i] = TYPE_CODES[t]
    return out


# ---------------------------------------------------------------------------------------------------
# C oracle wrappers
# ---------------------------------------------------------------------------------------------------
def set_time_grid(ttm: float, nb_steps_per_year: int = 360) -> Tuple[int, float]:
    n, dt = C.c_int(), C.c_double()
    lib().svo_set_time_grid(float(ttm), int(nb_steps_per_year), C.byref(n), C.byref(dt))
    return n.value, dt.value


def logsv_terminal_w(x0, sigma0, qvar0, dt, theta, kappa1, kappa2, beta, volvol, W0, W1,
                     eta=1.0, is_spot_measure=True):
    x, s, q = _state(x0, sigma0, qvar0)
    W0, W1 = _w(W0), _w(W1)
    nb_steps, n = W0.shape
    assert W1.shape == W0.shape and x.shape == s.shape == q.shape == (n,)
    lib().svo_logsv_terminal_w(n, nb_steps, dt, _p(x), _p(s), _p(q), theta, kappa1, kappa2, beta, volvol,
                               eta, int(bool(is_spot_measure)), _p(W0), _p(W1), n)
    return x, s, q


def heston_terminal_w(x0, var0, qvar0, dt, theta, kappa, rho, volvol, W0, W1):
    x, v, q = _state(x0, var0, qvar0)
    W0, W1 = _w(W0), _w(W1)
    nb_steps, n = W0.shape
    lib().svo_heston_terminal_w(n, nb_steps, dt, _p(x), _p(v), _p(q), theta, kappa, rho, volvol,
                                _p(W0), _p(W1), n)
    return x, v, q


def heston_qe_terminal_w(x0, var0, qvar0, dt, theta, kappa, rho, volvol, Z0, Z1, U):
    x, v, q = _state(x0, var0, qvar0)
    Z0, Z1, U = _w(Z0), _w(Z1), _w(U)
    nb_steps, n = Z0.shape
    lib().svo_heston_qe_terminal_w(n, nb_steps, dt, _p(x), _p(v), _p(q), theta, kappa, rho, volvol,
                                   _p(Z0), _p(Z1), _p(U), n)
    return x, v, q


def payoff(x, qvar, ttm, forward, strikes, optiontypes, discfactor=1.0, variable_type=LOG_RETURN):
    x = np.ascontiguousarray(x, dtype=np.float64)
    qvar = np.ascontiguousarray(qvar, dtype=np.float64)
    strikes = np.ascontiguousarray(strikes, dtype=np.float64)
    codes = type_codes(optiontypes) if not (isinstance(optiontypes, np.ndarray) and optiontypes.dtype == np.int8) \
        else optiontypes
    prices, stderrs = np.empty_like(strikes), np.empty_like(strikes)
    rc = lib().svo_payoff(x.size, _p(x), _p(qvar), ttm, forward, discfactor, strikes.size, _p(strikes),
                          codes.ctypes.data_as(C.POINTER(C.c_int8)), int(variable_type), _p(prices), _p(stderrs))
    if rc == -1:
        raise ValueError("unknown option payoff code")
    if rc == -2:
        raise NotImplementedError
    return prices, stderrs


def philox4x32_10(ctr: Sequence[int], key: Sequence[int]) -> Tuple[int, int, int, int]:
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().svo_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def normal_from_word(w: int) -> float:
    """one N(0,1) variate from one 32-bit word: the stream's piecewise-cubic inverse CDF (svo_normal_from_word)"""
    return float(lib().svo_normal_from_word(int(w) & 0xFFFFFFFF))


def fill_normals(seed, n_path, nb_steps, call_id=0, path_offset=0, step_offset=0, stream=0):
    W0 = np.empty((nb_steps, n_path)), np.empty((nb_steps, n_path))
    W0, W1 = W0
    lib().svo_fill_normals_stream(seed, call_id, stream, path_offset, step_offset, n_path, nb_steps, _p(W0), _p(W1),
                                  n_path)
    return W0, W1


def fill_uniforms(seed, n_path, nb_steps, call_id=0, path_offset=0, step_offset=0):
    U = np.empty((nb_steps, n_path))
    lib().svo_fill_uniforms(seed, call_id, path_offset, step_offset, n_path, nb_steps, _p(U), n_path)
    return U


def logsv_terminal_rng(x0, sigma0, qvar0, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, seed,
                       eta=1.0, is_spot_measure=True, call_id=0, path_offset=0, step_offset=0):
    x, s, q = _state(x0, sigma0, qvar0)
    lib().svo_logsv_terminal_rng(x.size, nb_steps, dt, _p(x), _p(s), _p(q), theta, kappa1, kappa2, beta, volvol,
                                 eta, int(bool(is_spot_measure)), seed, call_id, path_offset, step_offset)
    return x, s, q


def heston_terminal_rng(x0, var0, qvar0, nb_steps, dt, theta, kappa, rho, volvol, seed,
                        scheme=HESTON_EULER_FLOOR, call_id=0, path_offset=0, step_offset=0):
    x, v, q = _state(x0, var0, qvar0)
    lib().svo_heston_terminal_rng(x.size, nb_steps, dt, _p(x), _p(v), _p(q), theta, kappa, rho, volvol,
                                  int(scheme), seed, call_id, path_offset, step_offset)
    return x, v, q


def logsv_vol_paths(nb_steps, dt, v0, theta, kappa1, kappa2, beta, volvol, n_path, is_spot_measure=True,
                    brownians=None, seed=0, call_id=0, path_offset=0):
    out = np.empty((nb_steps + 1, n_path))
    if brownians is not None:
        brownians = _w(brownians)
        assert brownians.shape == (nb_steps, n_path)
    lib().svo_logsv_vol_paths(_p(out), n_path, n_path, nb_steps, dt, v0, theta, kappa1, kappa2, beta, volvol,
                              int(bool(is_spot_measure)), _p(brownians) if brownians is not None else None, n_path,
                              seed, call_id, path_offset)
    return out


# ---------------------------------------------------------------------------------------------------
# chain drivers (restating pricers/logsv_pricer.py:806-867, :1100-1162, pricers/heston_pricer.py:285-331)
# ---------------------------------------------------------------------------------------------------
def logsv_chain_fixed_randoms(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, W0s, W1s, dts,
                              v0, theta, kappa1, kappa2, beta, volvol, vol_backbone_etas,
                              is_spot_measure=True, variable_type=LOG_RETURN, return_states=False):
    n = np.asarray(W0s[0]).shape[1]
    x, s, q = np.zeros(n), v0 * np.ones(n), np.zeros(n)
    prices, stderrs, states = [], [], []
    for ttm, F, DF, K, ty, eta, W0, W1, dt in zip(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms,
                                                  vol_backbone_etas, W0s, W1s, dts):
        x, s, q = logsv_terminal_w(x, s, q, float(dt), theta, kappa1, kappa2, beta, volvol, W0, W1,
                                   eta=float(eta), is_spot_measure=is_spot_measure)
        p, e = payoff(x, q, float(ttm), float(F), K, ty, float(DF), variable_type)
        prices.append(p), stderrs.append(e), states.append((x.copy(), s.copy(), q.copy()))
    return (prices, stderrs, states) if return_states else (prices, stderrs)


# ---------------------------------------------------------------------------------------------------
# NumPy restatement (array code, one vector pass per line like the reference) -- second witness for
# the C oracle and the like-for-like single-core baseline of the reference's NumPy/Numba array code.
# ---------------------------------------------------------------------------------------------------
def np_logsv_terminal_w(x0, sigma0, qvar0, dt, theta, kappa1, kappa2, beta, volvol, W0, W1,
                        eta=1.0, is_spot_measure=True):
    """pricers/logsv_pricer.py:1028-1045 restated."""
    x, sigma, qvar = (np.array(a, dtype=np.float64) for a in (x0, sigma0, qvar0))
    sdt = np.sqrt(dt)
    alpha, adj = (-1.0, 0.0) if is_spot_measure else (1.0, beta * eta)
    vartheta2 = beta * beta + volvol * volvol
    eta2 = eta * eta
    L = np.log(sigma)
    for w0, w1 in zip(np.asarray(W0), np.asarray(W1)):
        w0 = sdt * w0
        w1 = sdt * w1
        s2dt = eta2 * sigma * sigma * dt
        x = x + alpha * 0.5 * s2dt + eta * sigma * w0
        L = L + ((kappa1 * theta / sigma - kappa1) + kappa2 * (theta - sigma) + adj * sigma - 0.5 * vartheta2) * dt \
            + beta * w0 + volvol * w1
        sigma = np.exp(L)
        qvar = qvar + 0.5 * (s2dt + eta2 * sigma * sigma * dt)
    return x, sigma, qvar


def np_heston_terminal_w(x0, var0, qvar0, dt, theta, kappa, rho, volvol, W0, W1):
    """pricers/heston_pricer.py:368-379 restated."""
    x, var, qvar = (np.array(a, dtype=np.float64) for a in (x0, var0, qvar0))
    sdt = np.sqrt(dt)
    rho_1 = np.sqrt(1.0 - rho * rho)
    for w0, w1 in zip(np.asarray(W0), np.asarray(W1)):
        w0 = sdt * w0
        w1 = sdt * w1
        sigma = np.sqrt(var)
        s2dt = var * dt
        x = x - 0.5 * s2dt + sigma * w0
        qvar = qvar + s2dt
        var = var + kappa * (theta - var) * dt + sigma * volvol * (rho * w0 + rho_1 * w1)
        var = np.maximum(var, 1e-4)
    return x, var, qvar


def np_payoff(x, qvar, ttm, forward, strikes, optiontypes, discfactor=1.0, variable_type=LOG_RETURN):
    """utils/mc_payoffs.py:61-88 restated."""
    x = np.asarray(x, dtype=np.float64)
    spots = forward * np.exp(x)
    spots = spots - (np.nanmean(spots) - forward)
    if variable_type == LOG_RETURN:
        u = spots
    elif variable_type == Q_VAR:
        u = np.asarray(qvar) / ttm
    else:
        raise NotImplementedError
    prices, stds = np.zeros(len(strikes)), np.zeros(len(strikes))
    with np.errstate(all="ignore"):
        for i, (K, ty) in enumerate(zip(strikes, optiontypes)):
            ty = str(ty)
            if ty == "C":
                pay = np.where(u > K, u - K, 0.0)
            elif ty == "IC":
                pay = np.where(u > K, u - K, 0.0) / spots
            elif ty == "P":
                pay = np.where(u < K, K - u, 0.0)
            elif ty == "IP":
                pay = np.where(u < K, K - u, 0.0) / spots
            else:
                raise ValueError("unknown option payoff code")
            prices[i] = discfactor * np.nanmean(pay)
            stds[i] = discfactor * np.nanstd(pay)
    return prices, stds / np.sqrt(x.shape[0])


# ---------------------------------------------------------------------------------------------------
# analytic side (oracle/svmc_oracle_analytic.c): affine-expansion MGF + Fourier inversion, config C5
# ---------------------------------------------------------------------------------------------------
def _cp(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def phi_grid(vol_scaler: float, is_spot_measure: bool = True, max_phi: int = 1000) -> np.ndarray:
    """utils/mgf_pricer.py:11-34"""
    p = np.linspace(0, 5.6 / vol_scaler, max_phi)
    return (-0.5 if is_spot_measure else 0.5) + 1j * p


def set_vol_scaler(sigma0: float, ttm: float) -> float:
    """pricers/logsv_pricer.py:664-666"""
    return sigma0 * np.sqrt(np.minimum(np.min(ttm), 0.5 / 12.0))


def logsv_mgf_grid(phi, psi, ttm, sigma0, theta, kappa1, kappa2, beta, volvol, a_t0=None, is_spot_measure=True,
                   expansion_order=2, vol_backbone_eta=1.0, rtol=1e-10, atol=1e-12):
    phi = np.ascontiguousarray(phi, dtype=np.complex128)
    psi = np.ascontiguousarray(psi, dtype=np.complex128)
    n = 5 if expansion_order == 2 else 3
    a = np.zeros((phi.size, n), dtype=np.complex128) if a_t0 is None else np.array(a_t0, dtype=np.complex128, order="C")
    lm = np.empty(phi.size, dtype=np.complex128)
    lib().svo_logsv_mgf_grid(phi.size, _cp(phi), _cp(psi), float(ttm), sigma0, theta, kappa1, kappa2, beta, volvol,
                             int(bool(is_spot_measure)), int(expansion_order), float(vol_backbone_eta), _cp(a), _cp(lm),
                             rtol, atol)
    return a, lm


def logsv_ode_rhs(phi, psi, A, theta, kappa1, kappa2, beta, volvol, is_spot_measure=True, expansion_order=2, eta=1.0):
    """A' of the affine-expansion coefficient ODE at one grid point (svmc_oracle_analytic.c ode_rhs)"""
    ph = np.array([complex(phi).real, complex(phi).imag])
    ps = np.array([complex(psi).real, complex(psi).imag])
    a = np.ascontiguousarray(np.asarray(A, dtype=np.complex128))
    out = np.empty(5, dtype=np.complex128)
    lib().svo_logsv_ode_rhs(theta, kappa1, kappa2, beta, volvol, int(bool(is_spot_measure)), int(expansion_order), eta,
                            _p(ph), _p(ps), _cp(a), _cp(out))
    return out


def heston_mgf_grid(phi, psi, ttm, v0, theta, kappa, volvol, rho, a_t0=None, b_t0=None):
    phi = np.ascontiguousarray(phi, dtype=np.complex128)
    psi = np.ascontiguousarray(psi, dtype=np.complex128)
    have = a_t0 is not None
    a = np.array(a_t0, dtype=np.complex128) if have else np.zeros(phi.size, dtype=np.complex128)
    b = np.array(b_t0, dtype=np.complex128) if have else np.zeros(phi.size, dtype=np.complex128)
    lm = np.empty(phi.size, dtype=np.complex128)
    lib().svo_heston_mgf_grid(phi.size, _cp(phi), _cp(psi), float(ttm), v0, theta, kappa, volvol, rho, _cp(a), _cp(b),
                              int(have), _cp(lm))
    return lm, a, b


def mgf_vanilla_slice(phi, log_mgf, forward, strikes, optiontypes, discfactor=1.0, is_spot_measure=True):
    phi = np.ascontiguousarray(phi, dtype=np.complex128)
    log_mgf = np.ascontiguousarray(log_mgf, dtype=np.complex128)
    strikes = np.ascontiguousarray(strikes, dtype=np.float64)
    codes = type_codes(optiontypes)
    prices = np.empty_like(strikes)
    rc = lib().svo_mgf_vanilla_slice(phi.size, _cp(phi), _cp(log_mgf), float(forward), strikes.size, _p(strikes),
                                     codes.ctypes.data_as(C.POINTER(C.c_int8)), float(discfactor),
                                     int(bool(is_spot_measure)), _p(prices))
    if rc != 0:
        raise ValueError("not implemented")
    return prices


def psi_grid() -> np.ndarray:
    """utils/mgf_pricer.py:37-47"""
    return -0.5 + 1j * np.linspace(0, 4000, 40000)


def mgf_qvar_slice(psi, log_mgf, ttm, strikes, optiontypes, discfactor=1.0):
    psi = np.ascontiguousarray(psi, dtype=np.complex128)
    log_mgf = np.ascontiguousarray(log_mgf, dtype=np.complex128)
    strikes = np.ascontiguousarray(strikes, dtype=np.float64)
    codes = type_codes(optiontypes)
    prices = np.empty_like(strikes)
    rc = lib().svo_mgf_qvar_slice(psi.size, _cp(psi), _cp(log_mgf), float(ttm), strikes.size, _p(strikes),
                                  codes.ctypes.data_as(C.POINTER(C.c_int8)), float(discfactor), _p(prices))
    if rc != 0:
        raise ValueError("not implemented")
    return prices


def logsv_chain_pricer(params, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, is_spot_measure=True,
                       expansion_order=2, vol_scaler=None, etas=None, rtol=1e-10, atol=1e-12, variable_type=LOG_RETURN):
    """pricers/logsv_pricer.py:669-739.  params = (sigma0, theta, kappa1, kappa2, beta, volvol)"""
    sigma0, theta, kappa1, kappa2, beta, volvol = params
    if vol_scaler is None:
        vol_scaler = set_vol_scaler(sigma0, np.min(ttms))
    if variable_type == LOG_RETURN:
        phi = phi_grid(vol_scaler, is_spot_measure)
        psi = np.zeros_like(phi)
    else:                                              # utils/mgf_pricer.py:77-83
        psi = psi_grid()
        phi = (np.zeros_like if is_spot_measure else np.ones_like)(psi)
    a, t0, out = None, 0.0, []
    for i, ttm in enumerate(ttms):
        a, lm = logsv_mgf_grid(phi, psi, ttm - t0, sigma0, theta, kappa1, kappa2, beta, volvol, a_t0=a,
                               is_spot_measure=is_spot_measure, expansion_order=expansion_order,
                               vol_backbone_eta=1.0 if etas is None else float(etas[i]), rtol=rtol, atol=atol)
        if variable_type == LOG_RETURN:
            out.append(mgf_vanilla_slice(phi, lm, forwards[i], strikes_ttms[i], optiontypes_ttms[i], discfactors[i],
                                         is_spot_measure))
        else:
            out.append(mgf_qvar_slice(psi, lm, ttm, strikes_ttms[i], optiontypes_ttms[i], discfactors[i]))
        t0 = ttm
    return out


def heston_chain_pricer(v0, theta, kappa, volvol, rho, ttms, forwards, strikes_ttms, optiontypes_ttms, discfactors,
                        vol_scaler=None, variable_type=LOG_RETURN):
    """pricers/heston_pricer.py:217-282 (LOG_RETURN on the phi grid, Q_VAR calls on the psi grid)"""
    if vol_scaler is None:
        vol_scaler = np.minimum(0.3, np.sqrt(v0 * ttms[0]))
    if variable_type == LOG_RETURN:
        phi = phi_grid(vol_scaler, True)
        psi = np.zeros_like(phi)
    else:
        psi = psi_grid()
        phi = np.zeros_like(psi)
    a = np.zeros(phi.size, dtype=np.complex128)
    b = np.zeros(phi.size, dtype=np.complex128)
    t0, out = 0.0, []
    for i, ttm in enumerate(ttms):
        lm, a, b = heston_mgf_grid(phi, psi, ttm - t0, v0, theta, kappa, volvol, rho, a_t0=a, b_t0=b)
        if variable_type == LOG_RETURN:
            out.append(mgf_vanilla_slice(phi, lm, forwards[i], strikes_ttms[i], optiontypes_ttms[i], discfactors[i]))
        else:
            out.append(mgf_qvar_slice(psi, lm, ttm, strikes_ttms[i], optiontypes_ttms[i], discfactors[i]))
        t0 = ttm
    return out


# ---------------------------------------------------------------------------------------------------
# rough LogSV (oracle/svmc_oracle_rough.c), SURVEY.md row f.4
# ---------------------------------------------------------------------------------------------------
def rough_randoms(ttms, nb_path, nb_steps_per_year=360, seed=10):
    """get_randoms_for_rough_vol_chain_valuation, pricers/logsv_pricer.py:1075-1097"""
    rng = np.random.RandomState(seed)
    grids = [np.linspace(0.0, t, int(t * nb_steps_per_year) + 2) for t in ttms]
    nb_last = grids[-1].size - 1
    Z0 = rng.normal(0, 1, size=(nb_last, nb_path))
    Z1 = rng.normal(0, 1, size=(nb_last, nb_path))
    return Z0, Z1, grids


def rough_logsv_chain_fixed_randoms(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, Z0, Z1, sigma0, theta,
                                    kappa1, kappa2, beta, orthog_vol, weights, nodes, timegrids, variable_type=LOG_RETURN,
                                    return_states=False):
    """rough_logsv_mc_chain_pricer_fixed_randoms, pricers/logsv_pricer.py:1164-1232: every expiry is simulated from
    time 0 on the first nb_steps rows of Z0/Z1"""
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    n = nodes.size
    v0 = np.full(n, sigma0 / np.sum(weights))
    volvol = np.sqrt(beta ** 2 + orthog_vol ** 2)
    rho = beta / volvol
    Z0, Z1 = _w(Z0), _w(Z1)
    nb_path = Z0.shape[1]
    prices, stderrs, states = [], [], []
    for ttm, F, DF, K, ty, grid in zip(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, timegrids):
        nb = grid.size - 1
        h = float(grid[1] - grid[0])
        ls, y = np.zeros(nb_path), np.zeros(nb_path)
        vol = np.ascontiguousarray(np.repeat(v0[:, None], nb_path, axis=1))
        lib().svo_rough_logsv_terminal_w(nb_path, nb, h, n, _p(nodes), _p(weights), _p(v0), theta, kappa1, kappa2, rho,
                                         volvol, _p(ls), _p(vol), _p(y), _p(Z0), _p(Z1), nb_path)
        p, e = payoff(ls, y, float(ttm), float(F), K, ty, float(DF), variable_type)
        # the reference hands compute_mc_vars_payoff a [1, nb_path] log-spot, so its "/ sqrt(x0.shape[0])"
        # (utils/mc_payoffs.py:88) divides by 1: the second return is the payoff's std, not its standard error
        e = e * np.sqrt(nb_path)
        prices.append(p), stderrs.append(e), states.append((ls, vol, y))
    return (prices, stderrs, states) if return_states else (prices, stderrs)
