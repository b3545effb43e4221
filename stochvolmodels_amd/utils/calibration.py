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
    optimizer step."""

    def __init__(self, model_vols: Callable[[np.ndarray], List[np.ndarray]], market_vols: np.ndarray,
                 weights: np.ndarray):
        self.model_vols = model_vols
        self.market_vols = np.asarray(market_vols, dtype=float)
        self.weights = np.asarray(weights, dtype=float)
        self.n_eval = 0

    def __call__(self, pars: np.ndarray, args=None) -> float:
        self.n_eval += 1
        vols = to_flat_np_array(self.model_vols(pars))
        return float(np.nansum(self.weights * np.square(vols - self.market_vols)))


def minimize_slsqp(objective: Callable, p0: np.ndarray, bounds: Sequence[Tuple[float, float]], constraints=None,
                   disp: bool = True, ftol: float = 1e-8) -> np.ndarray:
    """scipy SLSQP with the reference's options (ftol 1e-8, args=None), then the result check"""
    from scipy.optimize import minimize
    kwargs = dict(args=None, method="SLSQP", bounds=bounds, options={"disp": disp, "ftol": ftol})
    if constraints is not None:
        kwargs["constraints"] = constraints
    return validate_optimization_result(minimize(objective, p0, **kwargs), bounds)
