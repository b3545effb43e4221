"""
Transform grids, quadrature weights and the Fourier slice pricer (mirror of the reference's utils/mgf_pricer.py:
get_phi_grid :11-34, get_psi_grid :37-47, get_theta_grid :50-58, get_transform_var_grid :61-94,
compute_integration_weights :97-155, vanilla_slice_pricer_with_mgf_grid :174-221).  Grid construction is host
NumPy; the strike sums of the slice pricer run in libsvmc's mgf_vanilla_slice_kernel.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from ..analytic import AnalyticGrid, vanilla_prices_from_capped
from .config import VariableType


def get_phi_grid(is_spot_measure: bool = True, max_phi: int = 1000, vol_scaler: float = 0.28,
                 real_phi: float = None) -> np.ndarray:
    p = np.linspace(0, 5.6 / vol_scaler, max_phi)
    real_p = (-0.5 if is_spot_measure else 0.5) if real_phi is None else real_phi
    return real_p + 1j * p


def get_psi_grid() -> np.ndarray:
    return -0.5 + 1j * np.linspace(0, 4000, 40000)


def get_theta_grid() -> np.ndarray:
    return 0.0 + 1j * np.linspace(0, 600, 5000)


def get_transform_var_grid(variable_type: VariableType = VariableType.LOG_RETURN, is_spot_measure: bool = True,
                           max_phi: int = 1000, vol_scaler: float = 0.28, real_phi: float = None
                           ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    code = int(getattr(variable_type, "value", variable_type))
    if code == 1:
        phi_grid = get_phi_grid(is_spot_measure=is_spot_measure, max_phi=max_phi, vol_scaler=vol_scaler, real_phi=real_phi)
        psi_grid = np.zeros_like(phi_grid, dtype=np.complex128)
        theta_grid = np.zeros_like(phi_grid, dtype=np.complex128)
    elif code == 2:
        psi_grid = get_psi_grid()
        phi_grid = (np.zeros_like if is_spot_measure else np.ones_like)(psi_grid, dtype=np.complex128)
        theta_grid = np.zeros_like(phi_grid, dtype=np.complex128)
    elif code == 3:
        theta_grid = get_theta_grid()
        phi_grid = np.zeros_like(theta_grid, dtype=np.complex128)
        psi_grid = np.zeros_like(theta_grid, dtype=np.complex128)
    else:
        raise NotImplementedError
    return phi_grid, psi_grid, theta_grid


def compute_integration_weights(var_grid: np.ndarray, is_simpson: bool = True) -> np.ndarray:
    """validated composite Simpson / trapezoid weights on Im(var_grid) (reference :97-155)"""
    p = np.imag(var_grid)
    if len(p) < (3 if is_simpson else 2):
        raise ValueError("integration grid is too short for the selected rule")
    if not np.all(np.isfinite(p)):
        raise ValueError("integration grid must contain only finite values")
    steps = p[1:] - p[:-1]
    if np.any(steps <= 0.0):
        raise ValueError("integration grid must be strictly increasing")
    if np.any(np.abs(steps - steps[0]) > 1.0e-12 * max(1.0, np.abs(steps[0]))):
        raise ValueError("integration grid must be uniformly spaced")
    if is_simpson:
        if len(p) % 2 == 0:
            raise ValueError("Simpson integration requires an odd number of grid points")
        dp = 2.0 * np.ones(len(p))
        dp[0] = dp[-1] = 1.0
        dp[1::2] = 4.0
        return ((p[1] - p[0]) / 3.0) * dp
    dp = steps[0] * np.ones(len(p))
    dp[0] = dp[-1] = 0.5 * steps[0]
    return dp


def vanilla_slice_pricer_with_mgf_grid(log_mgf_grid: np.ndarray, phi_grid: np.ndarray, forward: float,
                                       strikes: np.ndarray, optiontypes: np.ndarray, discfactor: float = 1.0,
                                       is_spot_measure: bool = True, is_simpson: bool = True) -> np.ndarray:
    """vanilla prices of one slice from log E on the phi grid (reference :174-221), grids with |Re phi| = 1/2."""
    if not is_simpson or not np.all(np.abs(np.real(phi_grid)) == 0.5):
        raise NotImplementedError("the GPU slice pricer covers the Simpson rule on phi = +/-0.5 + i p grids")
    grid = AnalyticGrid(phi_grid, np.zeros_like(phi_grid), 1)
    try:
        lm = grid._up(np.ascontiguousarray(log_mgf_grid, dtype=np.complex128))
        capped = grid.capped_sums(forward, np.asarray(strikes, dtype=np.float64), log_mgf_ptr=lm.ptr)
        lm.free()
    finally:
        grid.close()
    return vanilla_prices_from_capped(capped, forward, strikes, optiontypes, discfactor, is_spot_measure)
