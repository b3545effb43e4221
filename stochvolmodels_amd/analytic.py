"""
Host side of the analytic (Fourier) chain pricers: device buffers for the transform grid and thin callers of
libsvmc's svmc_logsv_mgf_grid / svmc_heston_mgf_grid / svmc_mgf_vanilla_slice (csrc/svmc_analytic.hip).
Complex arrays travel as numpy.complex128 <-> interleaved doubles.  GPU only, like the Monte Carlo path.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .engine import DeviceBuffer, _current_device

# Dormand-Prince tolerances of the coefficient ODEs.  The reference uses SciPy's RK45 defaults (1e-3 / 1e-6),
# i.e. prices good to ~1e-6..1e-4; these reproduce the reference with its solver tightened to 1e-13 in price.
# The chain pricers take `ode_rtol=` / `ode_atol=` to trade that margin for time (tools/r04/analytic_tolerance_probe.py, a
# 4 x 21 chain, five parameter sets, DOP853: 1e-10 / 1e-12 -> 0.40-0.79 ms; 1e-8 / 1e-10 -> 0.31-0.71 ms, prices within
# 1.1e-11 of the default's; 1e-6 / 1e-8 -> 0.28-0.67 ms, within 4.5e-9; all of them stay 6.8e-7 from the reference as shipped
# -- its own solver's error.  With the 5(4) pair of rounds 1-3 the same settings took 0.92-1.22 / 0.52-0.84 / 0.39-0.69 ms).
ODE_RTOL, ODE_ATOL = 1e-10, 1e-12


# Grids are pooled per (thread, device, class, sizes): a chain pricing took a grid's five allocations, two uploads (each with its own
# wait) and five frees -- 58 us per chain (tools/r04/analytic_grid_probe.py) of an 0.9-1.2 ms pricing, and the whole of it
# again at every objective evaluation of an analytic calibration, whose transform grid never changes.  acquire() hands back
# the pooled object with its ODE state zeroed (queued memsets) and re-uploads phi / psi only when their bytes changed;
# release() returns it.  One pooled grid per key: a second acquire() before the release() builds a private one.
_POOL = {}
_POOL_LOCK = threading.Lock()
MAX_POOLED_GRIDS = 8                        # resident pooled grids per process (a 1000-point, 5-coefficient grid is ~130 KB of HBM)


class _Pooled:
    @classmethod
    def acquire(cls, *args):
        key = (threading.get_ident(), _current_device(), cls.__name__) + cls._pool_key(*args)
        with _POOL_LOCK:
            obj = _POOL.pop(key, None)
            if len(_POOL) > 8:                                  # grids of threads that ended: free their HBM
                alive = {t.ident for t in threading.enumerate()}
                for k in [k for k in _POOL if k[0] not in alive]:
                    _POOL.pop(k).close()
            while len(_POOL) > MAX_POOLED_GRIDS:                # ... and the least recently released ones beyond the cap: a
                _POOL.pop(next(iter(_POOL))).close()            # live thread that prices many grid shapes does not keep them all
        if obj is None:
            obj = cls(*args)
        else:
            obj._reset(*args)
        obj._pool_slot = key
        return obj

    def release(self) -> None:
        key = getattr(self, "_pool_slot", None)
        with _POOL_LOCK:
            if key is not None and key not in _POOL:
                _POOL[key] = self
                return
        self.close()


class AnalyticGrid(_Pooled):
    """transform grid phi (and psi) resident on the device, with the per-grid-point ODE state carried slice to slice"""

    @staticmethod
    def _pool_key(phi, psi, n_coef):
        return (int(np.asarray(phi).size), int(n_coef))

    def _reset(self, phi, psi, n_coef) -> None:
        # private copies: the pooled object compares the NEXT caller's grid with these -- a caller that reuses and mutates its
        # own array in place must not end up comparing the array with itself (and keeping a stale grid on the device)
        phi = np.array(phi, dtype=np.complex128, copy=True, order="C")
        psi = np.array(psi, dtype=np.complex128, copy=True, order="C")
        if not np.array_equal(phi, self.phi_host):
            _lib.check(self.lib.svmc_memcpy_h2d(self.phi.ptr, phi.ctypes.data, phi.nbytes, None))
            self.phi_host = phi
        if not np.array_equal(psi, self.psi_host):
            _lib.check(self.lib.svmc_memcpy_h2d(self.psi.ptr, psi.ctypes.data, psi.nbytes, None))
            self.psi_host = psi
        _lib.check(self.lib.svmc_stream_synchronize(None))     # pageable sources: the copies must not outlive the arrays
        _lib.check(self.lib.svmc_memset(self.a.ptr, 0, self.a.nbytes, None))
        _lib.check(self.lib.svmc_memset(self.b.ptr, 0, self.b.nbytes, None))

    def __init__(self, phi: np.ndarray, psi: np.ndarray, n_coef: int):
        self.lib = _lib.load()
        cnt = C.c_int()
        _lib.check(self.lib.svmc_device_count(C.byref(cnt)))
        if cnt.value < 1:
            raise _lib.SvmcError("no HIP device visible: the analytic pricers run on the GPU only")
        self.n = int(phi.size)
        self.n_coef = int(n_coef)
        self.phi_host = np.array(phi, dtype=np.complex128, copy=True, order="C")
        self.phi = self._up(self.phi_host)
        self.psi_host = np.array(psi, dtype=np.complex128, copy=True, order="C")
        self.psi = self._up(self.psi_host)
        self.last_given_up = 0
        self.a = DeviceBuffer(2 * self.n * self.n_coef)
        self.b = DeviceBuffer(2 * self.n)
        self.log_mgf = DeviceBuffer(2 * self.n)
        _lib.check(self.lib.svmc_memset(self.a.ptr, 0, self.a.nbytes, None))
        _lib.check(self.lib.svmc_memset(self.b.ptr, 0, self.b.nbytes, None))
        self._capped: Optional[DeviceBuffer] = None

    def _up(self, z: np.ndarray) -> DeviceBuffer:
        buf = DeviceBuffer(2 * z.size)
        _lib.check(self.lib.svmc_memcpy_h2d(buf.ptr, z.ctypes.data, z.nbytes, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        return buf

    def _down(self, buf: DeviceBuffer, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.complex128)
        _lib.check(self.lib.svmc_memcpy_d2h(out.ctypes.data, buf.ptr, out.nbytes, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        return out

    def set_a(self, a_t0: np.ndarray) -> None:
        a_t0 = np.ascontiguousarray(a_t0, dtype=np.complex128)
        assert a_t0.shape == (self.n, self.n_coef)
        _lib.check(self.lib.svmc_memcpy_h2d(self.a.ptr, a_t0.ctypes.data, a_t0.nbytes, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))

    def get_a(self) -> np.ndarray:
        return self._down(self.a, (self.n, self.n_coef))

    def get_log_mgf(self) -> np.ndarray:
        return self._down(self.log_mgf, (self.n,))

    def logsv_advance(self, ttm, sigma0, theta, kappa1, kappa2, beta, volvol, is_spot_measure, expansion_order,
                      vol_backbone_eta, rtol: Optional[float] = None, atol: Optional[float] = None) -> None:
        _lib.check(self.lib.svmc_logsv_mgf_grid(self.phi.ptr, self.psi.ptr, self.n, float(ttm), float(sigma0),
                                                float(theta), float(kappa1), float(kappa2), float(beta), float(volvol),
                                                int(bool(is_spot_measure)), int(expansion_order), float(vol_backbone_eta),
                                                self.a.ptr, self.log_mgf.ptr, ODE_RTOL if rtol is None else float(rtol),
                                                ODE_ATOL if atol is None else float(atol), None))

    def heston_advance(self, ttm, v0, theta, kappa, volvol, rho, have_t0: bool) -> None:
        _lib.check(self.lib.svmc_heston_mgf_grid(self.phi.ptr, self.psi.ptr, self.n, float(ttm), float(v0), float(theta),
                                                 float(kappa), float(volvol), float(rho), self.a.ptr, self.b.ptr,
                                                 int(bool(have_t0)), self.log_mgf.ptr, None))

    def capped_sums(self, forward: float, strikes: np.ndarray, log_mgf_ptr: Optional[int] = None) -> np.ndarray:
        strikes = np.ascontiguousarray(strikes, dtype=np.float64)
        k = strikes.size
        if self._capped is None or self._capped.n < k:
            self._capped = DeviceBuffer(max(k, 32))
        _lib.check(self.lib.svmc_mgf_vanilla_slice(self.phi.ptr, log_mgf_ptr or self.log_mgf.ptr, self.n, float(forward),
                                                   strikes.ctypes.data_as(C.POINTER(C.c_double)), k, self._capped.ptr, None))
        out = np.empty(k)
        _lib.check(self.lib.svmc_memcpy_d2h(out.ctypes.data, self._capped.ptr, 8 * k, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        return out

    # -- the same two reductions QUEUED into slices of one result buffer, downloaded once per chain: a chain's expiries are
    #    then launched back to back (advance, invert, advance, invert, ...) with no host round trip between them -- the wait
    #    for each expiry's sums cost a wake-up, the interpreter and a launch latency per expiry with the GPU idle (~50 us each)
    def reserve_results(self, n_doubles: int) -> None:
        if self._capped is None or self._capped.n < n_doubles:
            if self._capped is not None:
                self._capped.free()
            self._capped = DeviceBuffer(max(int(n_doubles), 32))

    def queue_capped_sums(self, forward: float, strikes: np.ndarray, offset: int) -> None:
        strikes = np.ascontiguousarray(strikes, dtype=np.float64)      # copied by the call (kernel arguments / pinned staging)
        _lib.check(self.lib.svmc_mgf_vanilla_slice(self.phi.ptr, self.log_mgf.ptr, self.n, float(forward),
                                                   strikes.ctypes.data_as(C.POINTER(C.c_double)), strikes.size,
                                                   self._capped.offset(offset), None))

    def queue_qvar_sums(self, ttm: float, strikes: np.ndarray, offset: int) -> None:
        strikes = np.ascontiguousarray(strikes, dtype=np.float64)
        _lib.check(self.lib.svmc_mgf_qvar_slice(self.psi.ptr, self.log_mgf.ptr, self.n, float(ttm),
                                                strikes.ctypes.data_as(C.POINTER(C.c_double)), strikes.size,
                                                self._capped.offset(offset), None))

    def download_results(self, n_doubles: int) -> np.ndarray:
        """the queued sums of the chain -- and, in the same wait, the last expiry's log-MGF: a grid point the ODE integrator
        GAVE UP on (step floor / try cap of csrc/svmc_analytic.hip) is NaN there and stays NaN for every later expiry, and the
        inversion drops it like the reference's nansum -- silently.  self.last_given_up counts them (0 for every sane set):
        the chain pricers warn and a calibrator can penalise the evaluation."""
        out = np.empty(int(n_doubles))
        lm = np.empty(self.n, dtype=np.complex128)
        _lib.check(self.lib.svmc_memcpy_d2h(out.ctypes.data, self._capped.ptr, 8 * int(n_doubles), None))
        _lib.check(self.lib.svmc_memcpy_d2h(lm.ctypes.data, self.log_mgf.ptr, lm.nbytes, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        self.last_given_up = int(np.count_nonzero(np.isnan(lm.real) | np.isnan(lm.imag)))
        return out

    def qvar_sums(self, ttm: float, strikes: np.ndarray) -> np.ndarray:
        strikes = np.ascontiguousarray(strikes, dtype=np.float64)
        k = strikes.size
        if self._capped is None or self._capped.n < k:
            self._capped = DeviceBuffer(max(k, 32))
        _lib.check(self.lib.svmc_mgf_qvar_slice(self.psi.ptr, self.log_mgf.ptr, self.n, float(ttm),
                                                strikes.ctypes.data_as(C.POINTER(C.c_double)), k, self._capped.ptr, None))
        out = np.empty(k)
        _lib.check(self.lib.svmc_memcpy_d2h(out.ctypes.data, self._capped.ptr, 8 * k, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        return out

    def close(self) -> None:
        for b in (self.phi, self.psi, self.a, self.b, self.log_mgf, self._capped):
            if b is not None:
                b.free()


class AnalyticGridBatch(_Pooled):
    """the transform grids of SEVERAL LogSV parameter sets resident on the device, advanced expiry by expiry in one
    launch per expiry (svmc_logsv_mgf_grid_batch) and inverted in one launch per expiry (svmc_mgf_vanilla_slice_batch):
    config C5's five sets, or the bumped parameter vectors of a finite-difference gradient, side by side.  Bit-identical
    to one AnalyticGrid per set."""

    @staticmethod
    def _pool_key(phis, psis, n_coef):
        return (len(phis), int(np.asarray(phis[0]).size), int(n_coef))

    def _reset(self, phis, psis, n_coef) -> None:
        phi = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.complex128) for p in phis]))
        psi = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.complex128) for p in psis]))
        for buf, z, name in ((self.phi, phi, "phi_host"), (self.psi, psi, "psi_host")):
            if not np.array_equal(z, getattr(self, name)):
                _lib.check(self.lib.svmc_memcpy_h2d(buf.ptr, z.ctypes.data, z.nbytes, None))
                setattr(self, name, z)
        _lib.check(self.lib.svmc_stream_synchronize(None))
        _lib.check(self.lib.svmc_memset(self.a.ptr, 0, self.a.nbytes, None))

    def __init__(self, phis: Sequence[np.ndarray], psis: Sequence[np.ndarray], n_coef: int):
        self.lib = _lib.load()
        self.n_sets = len(phis)
        self.n = int(np.asarray(phis[0]).size)
        self.n_coef = int(n_coef)
        phi = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.complex128) for p in phis]))       # np.stack copies: private
        psi = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.complex128) for p in psis]))
        assert phi.shape == psi.shape == (self.n_sets, self.n)
        self.last_given_up = np.zeros(self.n_sets, dtype=int)
        self.phi_host, self.psi_host = phi, psi
        self.phi, self.psi = DeviceBuffer(2 * phi.size), DeviceBuffer(2 * psi.size)
        for buf, z in ((self.phi, phi), (self.psi, psi)):
            _lib.check(self.lib.svmc_memcpy_h2d(buf.ptr, z.ctypes.data, z.nbytes, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        self.a = DeviceBuffer(2 * self.n_sets * self.n * self.n_coef)
        self.log_mgf = DeviceBuffer(2 * self.n_sets * self.n)
        _lib.check(self.lib.svmc_memset(self.a.ptr, 0, self.a.nbytes, None))
        self._capped: Optional[DeviceBuffer] = None

    def logsv_advance(self, ttm: float, params_rows: np.ndarray, is_spot_measure: bool, expansion_order: int,
                      rtol: Optional[float] = None, atol: Optional[float] = None) -> None:
        """params_rows [n_sets][8] = (sigma0, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta, 0)"""
        rows = np.ascontiguousarray(params_rows, dtype=np.float64)
        assert rows.shape == (self.n_sets, 8)
        _lib.check(self.lib.svmc_logsv_mgf_grid_batch(self.phi.ptr, self.psi.ptr, self.n, self.n_sets, float(ttm),
                                                      rows.ctypes.data_as(C.POINTER(C.c_double)), int(bool(is_spot_measure)),
                                                      int(expansion_order), self.a.ptr, self.log_mgf.ptr,
                                                      ODE_RTOL if rtol is None else float(rtol),
                                                      ODE_ATOL if atol is None else float(atol), None))

    def capped_sums(self, forward: float, strikes: np.ndarray) -> np.ndarray:
        """-> [n_sets][n_strikes]"""
        strikes = np.ascontiguousarray(strikes, dtype=np.float64)
        k = strikes.size
        if self._capped is None or self._capped.n < k * self.n_sets:
            self._capped = DeviceBuffer(max(k * self.n_sets, 32))
        _lib.check(self.lib.svmc_mgf_vanilla_slice_batch(self.phi.ptr, self.log_mgf.ptr, self.n, self.n_sets, float(forward),
                                                         strikes.ctypes.data_as(C.POINTER(C.c_double)), k, self._capped.ptr,
                                                         None))
        out = np.empty((self.n_sets, k))
        _lib.check(self.lib.svmc_memcpy_d2h(out.ctypes.data, self._capped.ptr, out.nbytes, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        return out

    def reserve_results(self, n_doubles: int) -> None:
        if self._capped is None or self._capped.n < n_doubles:
            if self._capped is not None:
                self._capped.free()
            self._capped = DeviceBuffer(max(int(n_doubles), 32))

    def queue_capped_sums(self, forward: float, strikes: np.ndarray, offset: int) -> None:
        """the [n_sets][n_strikes] sums of one expiry, queued into the result buffer at `offset` (AnalyticGrid.queue_capped_sums)"""
        strikes = np.ascontiguousarray(strikes, dtype=np.float64)
        _lib.check(self.lib.svmc_mgf_vanilla_slice_batch(self.phi.ptr, self.log_mgf.ptr, self.n, self.n_sets, float(forward),
                                                         strikes.ctypes.data_as(C.POINTER(C.c_double)), strikes.size,
                                                         self._capped.offset(offset), None))

    def download_results(self, n_doubles: int) -> np.ndarray:
        """as AnalyticGrid.download_results; self.last_given_up is an array, one count per parameter set"""
        out = np.empty(int(n_doubles))
        lm = np.empty((self.n_sets, self.n), dtype=np.complex128)
        _lib.check(self.lib.svmc_memcpy_d2h(out.ctypes.data, self._capped.ptr, 8 * int(n_doubles), None))
        _lib.check(self.lib.svmc_memcpy_d2h(lm.ctypes.data, self.log_mgf.ptr, lm.nbytes, None))
        _lib.check(self.lib.svmc_stream_synchronize(None))
        self.last_given_up = np.count_nonzero(np.isnan(lm.real) | np.isnan(lm.imag), axis=1).astype(int)
        return out

    def close(self) -> None:
        for b in (self.phi, self.psi, self.a, self.log_mgf, self._capped):
            if b is not None:
                b.free()


def vanilla_prices_from_capped(capped: np.ndarray, forward: float, strikes: np.ndarray, optiontypes: Sequence,
                               discfactor: float, is_spot_measure: bool) -> np.ndarray:
    """the payoff algebra of vanilla_slice_pricer_with_mgf_grid, reference utils/mgf_pricer.py:199-219"""
    strikes = np.asarray(strikes, dtype=np.float64)
    x = np.log(forward / strikes)
    prices = np.zeros_like(x)
    for idx, (xk, strike, type_, cap) in enumerate(zip(x, strikes, optiontypes, capped)):
        type_ = str(type_)
        if is_spot_measure:
            if type_ == "C":
                prices[idx] = discfactor * (forward - strike * cap)
            elif type_ == "P":
                prices[idx] = discfactor * (strike - strike * cap)
            else:
                raise ValueError("not implemented")
        else:
            if type_ in ("IC", "C"):
                prices[idx] = forward * discfactor * (1.0 - cap)
            elif type_ in ("IP", "P"):
                prices[idx] = forward * discfactor * (np.exp(-xk) - cap)
            else:
                raise ValueError("not implemented")
    return prices


def qvar_prices_from_sums(sums: np.ndarray, ttm: float, optiontypes: Sequence, discfactor: float) -> np.ndarray:
    """the payoff algebra of slice_qvar_pricer_with_a_grid, reference utils/mgf_pricer.py:343-356 (calls only)"""
    prices = np.zeros_like(sums)
    for idx, (s, type_) in enumerate(zip(sums, optiontypes)):
        if str(type_) != "C":
            raise ValueError("not implemented")
        prices[idx] = np.maximum(discfactor * s / ttm, 1e-10)
    return prices
