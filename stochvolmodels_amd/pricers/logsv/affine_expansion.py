"""
Affine expansion of the LogSV MGF on the GPU (mirror of the numerical path of the reference's
pricers/logsv/affine_expansion.py: ExpansionOrder :43-55, get_expansion_n :58-65, compute_logsv_a_mgf_grid :570-685).

The coefficient ODEs A' = A^T M A + L A + H (Eq. 4.14; matrices of Eqs. 4.17 / 4.25) are integrated by
libsvmc's logsv_mgf_grid_kernel, one 16-lane row per transform-grid point, with the Dormand-Prince 8(5,3) pair (DOP853) at
rtol 1e-10 (the reference: a Python loop of scipy.solve_ivp RK45 calls at rtol 1e-3).  The reference's two alternative
integrators of the SAME system are accepted as flags and answered by that one device integrator, with a one-time warning:
`is_stiff_solver=True` (reference :229-303: scipy BDF with the analytic Jacobian -- the adaptive explicit pair simply takes
more steps where the system stiffens, and counts the grid points it gives up on) and `is_analytic=True` (reference :306-384:
daily steps, the linear part by eigendecomposition, ten fixed-point sweeps for the quadratic part -- an approximation of the
solution the device integrates to 1e-10, so the caller gets the ODE's solution rather than that scheme's discretisation of it).
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


_FLAG_NOTES = {
    "is_stiff_solver": "is_stiff_solver=True: the coefficient ODEs are integrated on the GPU by the adaptive explicit Dormand-Prince "
                       "8(5,3) pair at rtol 1e-10, not by scipy's BDF; stiff grid points cost it more steps (those it gives up on "
                       "are counted and warned about) -- the flag changes nothing",
    "is_analytic": "is_analytic=True: the reference's semi-analytic scheme (daily steps, eigendecomposition of the linear part, ten "
                   "fixed-point sweeps; pricers/logsv/affine_expansion.py:306-384) approximates the solution of the coefficient "
                   "ODEs; this package integrates those ODEs on the GPU to rtol 1e-10 instead, so the prices are the numerical "
                   "route's (is_analytic=False), not that scheme's discretisation of them",
}
_FLAGS_WARNED = set()


def note_integrator_flags(is_stiff_solver: bool = False, is_analytic: bool = False) -> None:
    """the reference's two integrator switches are accepted; each is answered by the one device integrator, said once"""
    import warnings
    for name, on in (("is_stiff_solver", is_stiff_solver), ("is_analytic", is_analytic)):
        if on and name not in _FLAGS_WARNED:
            _FLAGS_WARNED.add(name)
            warnings.warn(_FLAG_NOTES[name], RuntimeWarning, stacklevel=3)


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
    note_integrator_flags(is_stiff_solver, is_analytic)
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
