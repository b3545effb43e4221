"""
Variance-swap strike from a strip of out-of-the-money options (reference utils/var_swap_pricer.py:8-56): the static
replication K_var^2 = (2/T) sum_i dK_i O(K_i)/K_i^2 - (F/K_atm - 1)^2 / T with O = put below the forward and call at
or above it, dK_i the centred strike spacing (one-sided at the ends) and K_atm the first strike at or above the forward.
"""
import numpy as np
import pandas as pd


def compute_var_swap_strike(puts: pd.Series, calls: pd.Series, forward: float, ttm: float) -> float:
    """puts / calls: undiscounted prices indexed by strike (a strike quoted on one side only leaves NaN on the
    other, which the sum skips).  Returns the strike as a VOLATILITY (square root of the annualised fair variance)."""
    strip = pd.concat([puts.rename("puts"), calls.rename("calls")], axis=1).sort_index()
    strikes = strip.index.to_numpy()
    below = strikes < forward
    dk = np.gradient(strikes)            # centred inside (0.5 (K[i+1] - K[i-1])), one-sided at the two ends
    otm = np.where(below, strip["puts"].to_numpy(), strip["calls"].to_numpy())
    fair = 2.0 * np.nansum(dk * otm / np.square(strikes))
    k_atm = strikes[~below][0]
    return float(np.sqrt((fair - np.square(forward / k_atm - 1.0)) / ttm))
