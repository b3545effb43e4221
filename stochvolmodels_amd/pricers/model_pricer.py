"""
ModelParams / ModelPricer: the Monte Carlo part of the reference's pricer interface
(pricers/model_pricer.py:28-41, :83-265).

Kept: price_chain, price_slice, price_vanilla, compute_chain_prices_with_vols, compute_model_ivols_for_chain, model_mc_price_chain,
simulate_terminal_values, simulate_vol_paths, compute_mc_chain_implied_vols, get_log_return_mc_pdf,
calibrate_model_params_to_chain with the reference's signatures.  Out of scope: the
matplotlib plotting methods and the slice / single-option conveniences built on them.
"""
from __future__ import annotations

from abc import ABC
from dataclasses import asdict, dataclass
from typing import List, Tuple

import numpy as np

from ..data.option_chain import OptionChain
from ..utils.calibration import CalibrationError, validate_optimization_result  # noqa: F401  (reference exports)
from ..utils.config import VariableType


@dataclass
class ModelParams:
    @classmethod
    def copy(cls, obj: "ModelParams") -> "ModelParams":
        return cls(**asdict(obj))


class ModelPricer(ABC):
    def __init__(self):
        super().__init__()

    def price_chain(self, option_chain: OptionChain, params: ModelParams, **kwargs) -> List[np.ndarray]:
        raise NotImplementedError("analytic chain pricing is outside the Monte Carlo hot path of this package")

    def compute_chain_prices_with_vols(self, option_chain: OptionChain, params: ModelParams,
                                       variable_type: VariableType = VariableType.LOG_RETURN, **kwargs
                                       ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """analytic chain prices and their Black implied vols (reference :109-120)"""
        model_prices = self.price_chain(option_chain=option_chain, params=params, variable_type=variable_type, **kwargs)
        return model_prices, option_chain.compute_model_ivols_from_chain_data(model_prices=model_prices)

    def compute_model_ivols_for_chain(self, option_chain: OptionChain, params: ModelParams, **kwargs
                                      ) -> List[np.ndarray]:
        return self.compute_chain_prices_with_vols(option_chain=option_chain, params=params, **kwargs)[1]

    def price_slice(self, params: ModelParams, ttm: float, forward: float, strikes: np.ndarray, optiontypes: np.ndarray,
                    discfactor: float = 1.0, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        """one expiry through the chain pricer: (prices, Black implied vols) (reference :156-178)"""
        chain = OptionChain.slice_to_chain(ttm=ttm, forward=forward, strikes=strikes, optiontypes=optiontypes,
                                           discfactor=discfactor)
        prices = self.price_chain(option_chain=chain, params=params, **kwargs)
        return prices[0], chain.compute_model_ivols_from_chain_data(model_prices=prices)[0]

    def price_vanilla(self, params: ModelParams, ttm: float, forward: float, strike: float, optiontype: str,
                      discfactor: float = 1.0, **kwargs) -> Tuple[float, float]:
        """one option through price_slice: (price, Black implied vol) (reference :180-195)"""
        prices, ivols = self.price_slice(params=params, ttm=ttm, forward=forward, strikes=np.array([strike]),
                                         optiontypes=np.array([optiontype]), discfactor=discfactor, **kwargs)
        return prices[0], ivols[0]

    def calibrate_model_params_to_chain(self, option_chain: OptionChain, **kwargs):
        raise NotImplementedError("must be implemented in parent class")

    def model_mc_price_chain(self, option_chain: OptionChain, params: ModelParams,
                             variable_type: VariableType = VariableType.LOG_RETURN, **kwargs
                             ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        raise NotImplementedError("must be implemented in parent class")

    def price_chain_with_mc(self, option_chain: OptionChain, params: ModelParams, **kwargs):
        """alias named by BASELINE.json; the reference's entry point is model_mc_price_chain."""
        return self.model_mc_price_chain(option_chain=option_chain, params=params, **kwargs)

    def simulate_vol_paths(self, params: ModelParams, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        raise NotImplementedError("must be implemented in parent class")

    def simulate_terminal_values(self, params: ModelParams, **kwargs) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        raise NotImplementedError("must be implemented in parent class")

    def compute_mc_chain_implied_vols(self, option_chain: OptionChain, params: ModelParams,
                                      variable_type: VariableType = VariableType.LOG_RETURN, nb_path: int = 100000,
                                      **kwargs) -> Tuple[List[np.ndarray], ...]:
        """MC prices +/- 1.96 stderr -> implied vols through the chain (reference :216-241).  The inversion
        itself is OptionChain.compute_model_ivols_from_chain_data (third-party in the reference)."""
        prices, stds = self.model_mc_price_chain(option_chain=option_chain, params=params,
                                                 variable_type=variable_type, nb_path=nb_path, **kwargs)
        std_factor = 1.96
        ups = [p + std_factor * s for p, s in zip(prices, stds)]
        downs = [np.maximum(p - std_factor * s, 1e-10) for p, s in zip(prices, stds)]
        ivols_mid = option_chain.compute_model_ivols_from_chain_data(model_prices=prices)
        ivols_up = option_chain.compute_model_ivols_from_chain_data(model_prices=ups)
        ivols_down = option_chain.compute_model_ivols_from_chain_data(model_prices=downs)
        return prices, ups, downs, ivols_mid, ivols_up, ivols_down, stds

    def get_log_return_mc_pdf(self, ttm: float, params: ModelParams, x_grid: np.ndarray, nb_path: int = 100000
                              ) -> np.ndarray:
        """Gaussian-kernel density of the simulated terminal log-returns on x_grid, normalised to sum to one
        (reference :243-265; contract: tests/test_model_calibration_contracts.py:121-139 -- paths that are NaN or
        beyond +-1e16 are counted, reported on stdout and left out).  simulate_terminal_values of the Monte Carlo
        pricers returns (x, vol, qvar): the density is that of x."""
        from scipy.stats import gaussian_kde
        sample = self.simulate_terminal_values(ttm=ttm, params=params, nb_path=nb_path)
        if isinstance(sample, (tuple, list)):
            sample = sample[0]
        sample = np.asarray(sample, dtype=np.float64).ravel()
        limit = 1e16
        is_nan = np.isnan(sample)
        too_high = ~is_nan & (sample > limit)
        too_low = ~is_nan & (sample < -limit)
        print(f"in mc: num -inf = {int(too_low.sum())}, num +inf = {int(too_high.sum())}, num nans = {int(is_nan.sum())}")
        density = gaussian_kde(sample[~(is_nan | too_high | too_low)])(np.asarray(x_grid, dtype=np.float64))
        return density / np.nansum(density)
