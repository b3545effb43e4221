"""
LogSvParams: parameters of the log-normal SV model with quadratic drift, Eq. (3.12)
    dsigma = (kappa1 + kappa2 sigma)(theta - sigma) dt + beta sigma dW0 + volvol sigma dW1
(the reference's pricers/logsv/logsv_params.py:34-161, plus the generator of the truncated volatility-moment
system :269-323 that the variance-swap backbone fit of the calibration uses).
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Any, Dict, Optional

import numpy as np
import pandas as pd

from ..model_pricer import ModelParams


@dataclass
class LogSvParams(ModelParams):
    sigma0: float = 0.2
    theta: float = 0.2
    kappa1: float = 1.0
    kappa2: Optional[float] = 2.5   # None maps to kappa1 / theta
    beta: float = -1.0
    volvol: float = 1.0
    vol_backbone: pd.Series = None
    H: float = 0.5
    weights: np.ndarray = None
    nodes: np.ndarray = None

    def __post_init__(self):
        if self.kappa2 is None:
            self.kappa2 = self.kappa1 / self.theta
        assert 1e-4 < self.H <= 0.5

    def approximate_kernel(self, T: float) -> None:
        """nodes / weights of the Markovian approximation of the rough kernel over the horizon T (reference :96-118):
        one node at 1e-3 with weight 1 for H in (0.49, 0.5] (the standard dynamics), otherwise the quadrature rule
        tuned for European options with 2 nodes for H in (0.4, 0.49] and 3 below (rough_logsv/rough_kernel.py)."""
        if 0.49 < self.H <= 0.5:
            self.weights = np.array([1.0])
            self.nodes = np.array([1e-3])
            return
        from ..rough_logsv.rough_kernel import european_rule
        self.nodes, self.weights = european_rule(self.H, 2 if 0.4 < self.H <= 0.49 else 3, T)

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    def to_str(self) -> str:
        return (f"sigma0={self.sigma0:0.2f}, theta={self.theta:0.2f}, kappa1={self.kappa1:0.2f}, "
                f"kappa2={self.kappa2:0.2f}, beta={self.beta:0.2f}, volvol={self.volvol:0.2f}")

    def set_vol_backbone(self, vol_backbone: pd.Series) -> None:
        self.vol_backbone = vol_backbone

    def _backbone_lookup(self, tau: float) -> float:
        # first quoted maturity at or beyond tau (reference utils/funcs.py:165-168, is_equal_or_largest)
        index = self.vol_backbone.index.to_numpy()
        return self.vol_backbone.loc[index[np.searchsorted(index, tau, side="left")]]

    def get_vol_backbone_eta(self, tau: float) -> float:
        return self._backbone_lookup(tau) if self.vol_backbone is not None else 1.0

    def get_vol_backbone_etas(self, ttms: np.ndarray) -> np.ndarray:
        etas = np.ones_like(ttms)
        if self.vol_backbone is not None:
            for idx, tau in enumerate(ttms):
                etas[idx] = self._backbone_lookup(tau)
        return etas

    @property
    def kappa(self) -> float:
        return self.kappa1 + self.kappa2 * self.theta

    @property
    def theta2(self) -> float:
        return self.theta * self.theta

    @property
    def vartheta2(self) -> float:
        return self.beta * self.beta + self.volvol * self.volvol

    @property
    def gamma(self) -> float:
        return self.kappa1 / self.theta

    def get_vol_moments_lambda(self, n_terms: int = 4) -> np.ndarray:
        """generator of the truncated moment system of Y = sigma - theta (reference :269-323; Eq. (3.48)).

        Ito on Y^n with dsigma = -(kappa + kappa2 Y) Y dt + vartheta (Y + theta) dW, kappa = kappa1 + kappa2 theta:
            d/dt E[Y^n] = (c_n - n kappa) E[Y^n] - n kappa2 E[Y^(n+1)] + 2 c_n theta E[Y^(n-1)] + c_n theta^2 E[Y^(n-2)],
            c_n = vartheta^2 n (n - 1) / 2.
        Row n of the matrix holds the coefficients of E[Y^(n-2)] .. E[Y^(n+1)] that fall on the unknowns
        E[Y^1] .. E[Y^n_terms]; the E[Y^0] = 1 term of row 2 and the closure of the last row go to the free vector
        (vol_moments_ode.compute_analytic_vol_moments)."""
        n = np.arange(1, n_terms + 1, dtype=float)
        c = 0.5 * self.vartheta2 * n * (n - 1.0)
        lam = np.diag(c - n * self.kappa)
        lam += np.diag(-n[:-1] * self.kappa2, k=1)
        lam += np.diag(2.0 * c[1:] * self.theta, k=-1)
        lam += np.diag(c[2:] * self.theta2, k=-2)
        return lam
