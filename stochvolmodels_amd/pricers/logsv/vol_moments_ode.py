"""
Moments of the LogSV volatility and the expected quadratic variance (Proposition 3.3 / Corollary 3.4 of Sepp and
Rakhmonov; reference pricers/logsv/vol_moments_ode.py:27-217), and the variance-swap fit of the vol backbone that the
PARAMS_WITH_VARSWAP_FIT calibration mode uses.  Host-side linear algebra on an n_terms x n_terms matrix (n_terms = 4
by default): not a kernel.

With Y = sigma - theta, M = (E[Y], .., E[Y^k]) solves dM/dt = Lambda M + C (LogSvParams.get_vol_moments_lambda),
C = (0, vartheta^2 theta^2, 0, .., -k kappa2 Y0^(k+1)) -- the last entry is the closure E[Y^(k+1)] ~ Y0^(k+1).  Then
    M(t)          = E M0 + P C,                 E = expm(Lambda t),  P = Lambda^-1 (E - I),
    int_0^t M ds  = P M0 + Lambda^-1 (P - t I) C,
both evaluated with linear solves instead of an explicit inverse.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy import linalg as sla

from .logsv_params import LogSvParams


def compute_analytic_vol_moments(params: LogSvParams, t: float = 1.0, n_terms: int = 4, is_qvar: bool = False
                                 ) -> np.ndarray:
    """E[Y_t^n], n = 1..n_terms (reference :27-108), or their integrals over [0, t] when is_qvar"""
    y = params.sigma0 - params.theta
    m0 = np.power(y, np.arange(1, n_terms + 1, dtype=float))
    if np.isclose(np.abs(t), 0.0):
        return m0
    lam = params.get_vol_moments_lambda(n_terms=n_terms)
    free = np.zeros(n_terms)
    free[1] = params.vartheta2 * params.theta2
    free[-1] = -n_terms * params.kappa2 * np.power(y, n_terms + 1)        # n_terms >= 3, as the reference
    e = sla.expm(lam * t)
    eye = np.eye(n_terms)
    p = np.linalg.solve(lam, e - eye)
    if is_qvar:
        return p @ m0 + np.linalg.solve(lam, p - t * eye) @ free
    return e @ m0 + p @ free


def compute_analytic_qvar(params: LogSvParams, ttm: float = 1.0, n_terms: int = 4) -> float:
    """annualised expected quadratic variance E[(1/T) int sigma^2 dt], Eq. (3.53) (reference :111-149)"""
    if np.isclose(ttm, 0.0):
        return float(np.square(params.sigma0))
    im = compute_analytic_vol_moments(params=params, t=ttm, n_terms=n_terms, is_qvar=True)
    return float((im[1] + 2.0 * params.theta * im[0]) / ttm + params.theta2)


def compute_vol_moments_t(params: LogSvParams, ttm: np.ndarray, n_terms: int = 4, is_print: bool = False) -> np.ndarray:
    out = np.zeros((len(ttm), n_terms))
    for idx, t_ in enumerate(ttm):
        out[idx, :] = compute_analytic_vol_moments(t=t_, params=params, n_terms=n_terms)
        if is_print:
            print(f"t={t_}: {out[idx, :]}")
    return out


def compute_expected_vol_t(params: LogSvParams, t: np.ndarray, n_terms: int = 4) -> np.ndarray:
    """E[sigma_t] over an array of maturities (reference :166-176)"""
    return np.array([compute_analytic_vol_moments(t=t_, params=params, n_terms=n_terms)[0] + params.theta for t_ in t])


def compute_sqrt_qvar_t(params: LogSvParams, t: np.ndarray, n_terms: int = 4) -> np.ndarray:
    """the model variance-swap rate sqrt(E[qvar]) over an array of maturities (reference :179-184)"""
    return np.array([np.sqrt(compute_analytic_qvar(ttm=t_, params=params, n_terms=n_terms)) for t_ in t])


def fit_model_vol_backbone_to_varswaps(log_sv_params: LogSvParams, varswap_strikes: pd.Series, n_terms: int = 4,
                                       verbose: bool = False) -> pd.Series:
    """backbone multipliers eta(T_i) that make the model reproduce the market's forward quadratic variance between
    quoted maturities (reference :186-217): eta_i = d(market T K_var^2) / d(model T qvar) over (T_{i-1}, T_i];
    non-positive ratios fall back to 1, and maturities under 0.06y take the square root (the reference's ad-hoc
    damping of the front end)."""
    ttms = varswap_strikes.index.to_numpy()
    market = ttms * np.square(varswap_strikes.to_numpy())
    model = ttms * np.array([compute_analytic_qvar(params=log_sv_params, ttm=ttm, n_terms=n_terms) for ttm in ttms])
    eta = np.diff(market, prepend=0.0) / np.diff(model, prepend=0.0)
    eta = np.where(eta > 0.0, eta, 1.0)
    eta = np.where(ttms < 0.06, np.sqrt(eta), eta)
    if verbose:
        print(pd.DataFrame({"varswap strike": varswap_strikes.to_numpy(), "market_qvar_dt": market,
                            "model_qvar_dt": model, "model_eta": eta}, index=ttms))
    return pd.Series(eta, index=ttms)
