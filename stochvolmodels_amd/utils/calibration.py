"""
Calibration plumbing shared by the pricers: the weighted implied-vol objective of Eq. (6.3), the
SLSQP call and the result check.  Mirrors what the reference does inside LogSVPricer / HestonPricer
.calibrate_model_params_to_chain (pricers/logsv_pricer.py:207-332, :440-557; pricers/heston_pricer.py:110-181;
validate_optimization_result pricers/model_pricer.py:48-80).

Everything here is host glue around GPU pricers: per objective evaluation the device prices the whole chain (MC on
resident fixed randoms, or the analytic transform grid) and hands back a few dozen prices; the Black inversion of
those and the weighted sum of squares stay on the host (tens of numbers -- a kernel launch would cost more than the
arithmetic).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

from .funcs import to_flat_np_array


class CalibrationError(RuntimeError):
    """the optimizer did not produce a usable parameter vector"""


def validate_optimization_result(result, bounds) -> np.ndarray:
    """the optimizer's vector if the run succeeded and the vector is numeric, 1-d, of len(bounds), finite and inside
    the box (1e-10 slack); CalibrationError with the optimizer's message otherwise (same contract and wording as the
    reference's pricers/model_pricer.py:48-80)"""
    message = str(getattr(result, "message", "no optimizer message"))

    def reject(what: str, cause=None):
        raise CalibrationError(f"{what}: {message}") from cause

    if not getattr(result, "success", False):
        reject("Calibration failed")
    if getattr(result, "x", None) is None:
        reject("Calibration returned no parameter vector")
    try:
        values = np.array(result.x, dtype=float)
    except (TypeError, ValueError) as error:
        reject("Calibration returned a non-numeric parameter vector", error)
    if values.shape != (len(bounds),):
        reject("Calibration returned a parameter vector with the wrong shape")
    if not np.isfinite(values).all():
        reject("Calibration returned non-finite parameters")
    slack = 1.0e-10
    lower = np.array([-np.inf if lo is None else lo for lo, _ in bounds], dtype=float)
    upper = np.array([np.inf if hi is None else hi for _, hi in bounds], dtype=float)
    if (values < lower - slack).any():
        reject("Calibration returned parameters below bounds")
    if (values > upper + slack).any():
        reject("Calibration returned parameters above bounds")
    return values


def chain_calibration_weights(option_chain, market_vols: np.ndarray, is_vega_weighted: bool, is_unit_ttm_vega: bool
                              ) -> np.ndarray:
    """flattened objective weights: per-slice normalised Black vegas, or ones"""
    if not is_vega_weighted:
        return np.ones_like(market_vols)
    vegas = option_chain.get_chain_vegas(is_unit_ttm_vega=is_unit_ttm_vega)
    return to_flat_np_array([v / sum(v) for v in vegas])


class ImpliedVolObjective:
    """pars -> sum_n w_n (sigma_model_n - sigma_market_n)^2 with NaN terms dropped (np.nansum).  `model_vols(pars)`
    returns the per-slice model implied vols.  Counts evaluations (`n_eval`) so callers can report kernel time per
    optimizer step.

    `model_vols_batch(list of pars)` (optional) returns the vols of SEVERAL parameter vectors from one batch of launches;
    `gradient` then hands SLSQP the forward-difference gradient it would otherwise build itself from n + 1 separate
    objective calls -- the same evaluation points (scipy.optimize approx_derivative, '2-point', abs_step = sqrt(eps),
    a step that would leave the box flipped), the same numbers, one launch batch per optimizer iterate."""

    FD_STEP = float(np.sqrt(np.finfo(float).eps))          # scipy's SLSQP default `eps`

    def __init__(self, model_vols: Callable[[np.ndarray], List[np.ndarray]], market_vols: np.ndarray,
                 weights: np.ndarray, model_vols_batch: Callable[[List[np.ndarray]], List[List[np.ndarray]]] = None,
                 bounds: Sequence[Tuple[float, float]] = None):
        self.model_vols = model_vols
        self.model_vols_batch = model_vols_batch
        self.market_vols = np.asarray(market_vols, dtype=float)
        self.weights = np.asarray(weights, dtype=float)
        self.n_eval = 0
        self.n_batches = 0
        self._last = (None, None)                          # (parameter bytes, objective value) of the latest call
        if bounds is not None:
            self.lower = np.array([-np.inf if lo is None else lo for lo, _ in bounds], dtype=float)
            self.upper = np.array([np.inf if hi is None else hi for _, hi in bounds], dtype=float)
        else:
            self.lower = self.upper = None

    def _value(self, vols) -> float:
        return float(np.nansum(self.weights * np.square(to_flat_np_array(vols) - self.market_vols)))

    def __call__(self, pars: np.ndarray, args=None) -> float:
        self.n_eval += 1
        value = self._value(self.model_vols(pars))
        self._last = (np.asarray(pars, dtype=float).tobytes(), value)
        return value

    def fd_steps(self, x0: np.ndarray) -> np.ndarray:
        """the signed absolute steps of the forward difference at x0 (approx_derivative's rule for a bounded box)"""
        h = np.full(x0.shape, self.FD_STEP)
        if self.lower is None or np.all(np.isinf(self.lower) & np.isinf(self.upper)):
            return h
        lower_dist, upper_dist = x0 - self.lower, self.upper - x0
        x = x0 + h
        violated = (x < self.lower) | (x > self.upper)
        fitting = np.abs(h) <= np.maximum(lower_dist, upper_dist)
        h[violated & fitting] *= -1.0
        forward = (upper_dist >= lower_dist) & ~fitting
        h[forward] = upper_dist[forward]
        backward = (upper_dist < lower_dist) & ~fitting
        h[backward] = -lower_dist[backward]
        return h

    def gradient(self, pars: np.ndarray, args=None) -> np.ndarray:
        x0 = np.asarray(pars, dtype=float)
        h = self.fd_steps(x0)
        points = []
        for i in range(x0.size):
            xi = x0.copy()
            xi[i] += h[i]
            points.append(xi)
        key, f0 = self._last
        need_f0 = key != x0.tobytes()
        if need_f0:
            points.append(x0.copy())
        vols = self.model_vols_batch(points)
        self.n_eval += len(points)
        self.n_batches += 1
        values = [self._value(v) for v in vols]
        if need_f0:
            f0 = values.pop()
        return np.array([(values[i] - f0) / (points[i][i] - x0[i]) for i in range(x0.size)])


def minimize_slsqp(objective: Callable, p0: np.ndarray, bounds: Sequence[Tuple[float, float]], constraints=None,
                   disp: bool = True, ftol: float = 1e-8, jac: Callable = None) -> np.ndarray:
    """scipy SLSQP with the reference's options (ftol 1e-8, args=None), then the result check.  `jac`: the objective's
    gradient when the caller can produce it in one batch (ImpliedVolObjective.gradient); SLSQP differences otherwise."""
    from scipy.optimize import minimize
    kwargs = dict(args=None, method="SLSQP", bounds=bounds, options={"disp": disp, "ftol": ftol})
    if jac is not None:
        kwargs["jac"] = jac
    if constraints is not None:
        kwargs["constraints"] = constraints
    return validate_optimization_result(minimize(objective, p0, **kwargs), bounds)
