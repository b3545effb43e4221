"""
compute_mc_vars_payoff on the GPU (mirror of the reference's utils/mc_payoffs.py:10-88).

Same signature, same return, same errors.  The path vectors are uploaded, reduced by libsvmc's two
deterministic sum kernels (svmc_spot_sums, svmc_payoff_sums) and finalised on the host; K prices come
back.  Inside the chain pricers the state is already resident and this function is not on the path.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from ..engine import get_engine, option_type_codes, payoff_finalize, payoff_shifts
from ..mc_chain import variable_type_code
from .config import VariableType


def compute_mc_vars_payoff(x0: np.ndarray, sigma0: np.ndarray, qvar0: np.ndarray, ttm: float, forward: float,
                           strikes_ttm: np.ndarray, optiontypes_ttm: np.ndarray, discfactor: float = 1.0,
                           variable_type: VariableType = VariableType.LOG_RETURN
                           ) -> Tuple[np.ndarray, np.ndarray]:
    """sigma0 is accepted for signature symmetry and unused, as in the reference."""
    vt = variable_type_code(variable_type)                       # NotImplementedError for SIGMA
    codes = option_type_codes(optiontypes_ttm)                   # ValueError("unknown option payoff code")
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    strikes = np.ascontiguousarray(strikes_ttm, dtype=np.float64)
    n = x0.shape[0]
    eng = get_engine(n)
    eng.upload(eng.x.ptr, x0)
    if vt == 2:
        eng.upload(eng.qvar.ptr, np.ascontiguousarray(qvar0, dtype=np.float64))
    shifts = payoff_shifts(strikes.ravel(), codes, float(forward), vt)
    ptr, _ = eng.alloc_sums(2 + 3 * strikes.size, "slice")
    eng.spot_sums(eng.x.ptr, float(forward), ptr)
    eng.payoff_sums(eng.x.ptr, eng.qvar.ptr if vt == 2 else None, float(forward), float(ttm), ptr, strikes.ravel(),
                    codes, shifts, vt, ptr + 16)
    sums = eng.download(ptr + 16, 3 * strikes.size)
    prices, stderrs = payoff_finalize(sums, shifts, float(discfactor), float(n))
    return prices.reshape(strikes.shape), stderrs.reshape(strikes.shape)
