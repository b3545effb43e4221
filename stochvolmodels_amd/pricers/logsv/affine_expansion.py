"""
Affine expansion of the LogSV MGF on the GPU (mirror of the numerical path of the reference's
pricers/logsv/affine_expansion.py: ExpansionOrder :43-55, get_expansion_n :58-65, compute_logsv_a_mgf_grid :570-685).

The coefficient ODEs A' = A^T M A + L A + H (Eq. 4.14; matrices of Eqs. 4.17 / 4.25) are integrated by
libsvmc's logsv_mgf_grid_kernel, one 16-lane row per transform-grid point, with the Dormand-Prince 8(5,3) pair (DOP853) at
rtol 1e-10 (the reference: a Python loop of scipy.solve_ivp RK45 calls at rtol 1e-3).  The semi-analytic
fixed-point path (`is_analytic=True`) and the BDF switch are not reproduced: `is_stiff_solver` is accepted and
ignored (the adaptive explicit pair simply takes more steps), `is_analytic=True` raises.
"""
from __future__ import annotations

from enum import Enum
from typing import Optional, Tuple

import numpy as np

from ...analytic import AnalyticGrid
from ...utils.config import VariableType


class ExpansionOrder(Enum):
    ZERO = 0
    FIRST = 1
    SECOND = 2


def get_expansion_n(expansion_order: ExpansionOrder = ExpansionOrder.FIRST) -> int:
    return 3 if expansion_order == ExpansionOrder.FIRST else 5


def _order_code(expansion_order) -> int:
    code = int(getattr(expansion_order, "value", expansion_order))
    if code not in (1, 2):
        raise NotImplementedError
    return code


def compute_logsv_a_mgf_grid(ttm: float, phi_grid: np.ndarray, psi_grid: np.ndarray, theta_grid: np.ndarray,
                             sigma0: float, theta: float, kappa1: float, kappa2: float, beta: float, volvol: float,
                             variable_type: VariableType = VariableType.LOG_RETURN,
                             expansion_order: ExpansionOrder = ExpansionOrder.SECOND,
                             a_t0: Optional[np.ndarray] = None, is_stiff_solver: bool = False,
                             is_analytic: bool = False, is_spot_measure: bool = True, vol_backbone_eta: float = 1.0,
                             **kwargs) -> Tuple[np.ndarray, np.ndarray]:
    """(A(ttm), log E) over the grid, from A(0) = a_t0 (zeros; -Theta in the 2nd slot for VariableType.SIGMA)."""
    if is_analytic:
        raise NotImplementedError("the semi-analytic fixed-point path is not part of this package")
    order = _order_code(expansion_order)
    n = 5 if order == 2 else 3
    if a_t0 is None:
        a_t0 = np.zeros((np.asarray(phi_grid).shape[0], n), dtype=np.complex128)
        if int(getattr(variable_type, "value", variable_type)) == 3:
            a_t0[:, 1] = -np.asarray(theta_grid)
    grid = AnalyticGrid(np.asarray(phi_grid), np.asarray(psi_grid), n)
    try:
        grid.set_a(a_t0)
        grid.logsv_advance(ttm, sigma0, theta, kappa1, kappa2, beta, volvol, is_spot_measure, order, vol_backbone_eta)
        return grid.get_a(), grid.get_log_mgf()
    finally:
        grid.close()
