"""
OptionChain: the validated chain container the Monte Carlo pricers consume
(hot-path subset of the reference's data/option_chain.py:126-246, :355-395, :462-492).

Out of scope here: implied-vol / vega / delta / varswap analytics, which call
the un-vendored `vanilla_option_pricers` package, and the SwOptionChain / FutOptionChain containers.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple, List, Optional, Sequence

import numpy as np
import pandas as pd

_VALID_OPTION_TYPES = frozenset({"C", "P", "IC", "IP"})


def _check_quotes(name: str, values, size: int, *, positive: bool = False, nonnegative: bool = False) -> np.ndarray:
    a = np.asarray(values)
    if a.ndim != 1 or a.size != size:
        raise ValueError(f"{name} must be one-dimensional with length {size}")
    if not np.all(np.isfinite(a)):
        raise ValueError(f"{name} must contain only finite values")
    if positive and np.any(a <= 0.0):
        raise ValueError(f"{name} must contain only positive values")
    if nonnegative and np.any(a < 0.0):
        raise ValueError(f"{name} must contain only non-negative values")
    return a


def _check_slice(strikes, optiontypes) -> int:
    k = np.asarray(strikes)
    t = np.asarray(optiontypes)
    if k.ndim != 1 or t.ndim != 1:
        raise ValueError("strikes and optiontypes must be one-dimensional")
    if k.size == 0:
        raise ValueError("strikes and optiontypes must not be empty")
    if k.size != t.size:
        raise ValueError("strikes and optiontypes must have the same length")
    if not np.all(np.isfinite(k)) or np.any(k <= 0.0):
        raise ValueError("strikes must contain only finite positive values")
    bad = set(t.astype(str)) - _VALID_OPTION_TYPES
    if bad:
        raise ValueError(f"unsupported optiontypes: {sorted(bad)}")
    return k.size


@dataclass
class OptionChain:
    ttms: np.ndarray
    forwards: np.ndarray
    strikes_ttms: Sequence[np.ndarray]
    optiontypes_ttms: Sequence[np.ndarray]
    ids: Optional[np.ndarray]
    discfactors: Optional[np.ndarray] = None
    discount_rates: Optional[np.ndarray] = None
    ticker: Optional[str] = None
    bid_ivs: Optional[Sequence[np.ndarray]] = None
    ask_ivs: Optional[Sequence[np.ndarray]] = None
    bid_prices: Optional[Sequence[np.ndarray]] = None
    ask_prices: Optional[Sequence[np.ndarray]] = None
    forwards0: Optional[np.ndarray] = None

    def __post_init__(self):
        ttms = np.asarray(self.ttms)
        forwards = np.asarray(self.forwards)
        if ttms.ndim != 1 or ttms.size == 0:
            raise ValueError("ttms must be a non-empty one-dimensional array")
        if not np.all(np.isfinite(ttms)) or np.any(ttms <= 0.0):
            raise ValueError("ttms must contain only finite positive values")
        if np.any(np.diff(ttms) <= 0.0):
            raise ValueError("ttms must be strictly increasing")
        if forwards.ndim != 1 or forwards.size != ttms.size:
            raise ValueError("ttms and forwards must have the same one-dimensional length")
        if not np.all(np.isfinite(forwards)) or np.any(forwards <= 0.0):
            raise ValueError("forwards must contain only finite positive values")
        if len(self.strikes_ttms) != ttms.size or len(self.optiontypes_ttms) != ttms.size:
            raise ValueError("ttms, strikes_ttms, and optiontypes_ttms must have the same length")
        if self.ids is not None and len(self.ids) != ttms.size:
            raise ValueError("ids and ttms must have the same length")

        if self.discfactors is not None:
            _check_quotes("discfactors", self.discfactors, ttms.size, positive=True)
            self.discount_rates = -np.log(self.discfactors) / self.ttms
        elif self.discount_rates is not None:
            _check_quotes("discount_rates", self.discount_rates, ttms.size)
            self.discfactors = np.exp(-self.discount_rates * self.ttms)
        else:
            self.discfactors = np.ones_like(self.ttms)
            self.discount_rates = np.zeros_like(self.ttms)

        if self.forwards0 is not None:
            _check_quotes("forwards0", self.forwards0, ttms.size, positive=True)

        quotes = (("bid_ivs", self.bid_ivs, True, False), ("ask_ivs", self.ask_ivs, True, False),
                  ("bid_prices", self.bid_prices, False, True), ("ask_prices", self.ask_prices, False, True))
        for name, values, _, _ in quotes:
            if values is not None and len(values) != ttms.size:
                raise ValueError(f"{name} and ttms must have the same length")
        for idx, (strikes, optiontypes) in enumerate(zip(self.strikes_ttms, self.optiontypes_ttms)):
            size = _check_slice(strikes, optiontypes)
            seen = {}
            for name, values, positive, nonnegative in quotes:
                if values is not None:
                    seen[name] = _check_quotes(f"{name}[{idx}]", values[idx], size, positive=positive,
                                               nonnegative=nonnegative)
            if "bid_ivs" in seen and "ask_ivs" in seen and np.any(seen["bid_ivs"] > seen["ask_ivs"]):
                raise ValueError(f"bid_ivs[{idx}] must not exceed ask_ivs[{idx}]")
            if "bid_prices" in seen and "ask_prices" in seen and np.any(seen["bid_prices"] > seen["ask_prices"]):
                raise ValueError(f"bid_prices[{idx}] must not exceed ask_prices[{idx}]")

    @classmethod
    def slice_to_chain(cls, ttm: float, forward: float, strikes: np.ndarray, optiontypes: np.ndarray,
                       discfactor: float = 1.0, id: Optional[str] = None) -> "OptionChain":
        return cls(ttms=np.array([ttm]), forwards=np.array([forward]), strikes_ttms=(strikes,),
                   optiontypes_ttms=(optiontypes,), discfactors=np.array([discfactor]),
                   ids=np.array([id]) if id is not None else np.array([f"{ttm:0.2f}"]))

    def get_mid_vols(self) -> Optional[List[np.ndarray]]:
        if self.bid_ivs is not None and self.ask_ivs is not None:
            return [0.5 * (b + a) for b, a in zip(self.bid_ivs, self.ask_ivs)]
        return None

    @classmethod
    def to_forward_normalised_strikes(cls, obj: "OptionChain") -> "OptionChain":
        return cls(ttms=obj.ttms, forwards=np.ones_like(obj.forwards),
                   strikes_ttms=[k / f for k, f in zip(obj.strikes_ttms, obj.forwards)],
                   optiontypes_ttms=obj.optiontypes_ttms, discfactors=obj.discfactors, ticker=obj.ticker,
                   ids=obj.ids, bid_ivs=obj.bid_ivs, ask_ivs=obj.ask_ivs, forwards0=obj.forwards)

    @classmethod
    def to_uniform_strikes(cls, obj: "OptionChain", num_strikes: int = 21) -> "OptionChain":
        strikes, types = [], []
        for k, forward in zip(obj.strikes_ttms, obj.forwards):
            grid = np.linspace(k[0], k[-1], num_strikes)
            strikes.append(grid)
            types.append(np.where(grid >= forward, "C", "P"))
        return cls(ttms=obj.ttms, forwards=obj.forwards, strikes_ttms=strikes, optiontypes_ttms=types,
                   discfactors=obj.discfactors, ticker=obj.ticker, ids=obj.ids, bid_ivs=None, ask_ivs=None)

    @classmethod
    def get_uniform_chain(cls, ttms: np.ndarray = np.array([0.083, 0.25]), ids: np.ndarray = np.array(["1m", "3m"]),
                          forwards: np.ndarray = np.array([1.0, 1.0]), strikes: np.ndarray = np.linspace(0.9, 1.1, 3),
                          flat_vol: float = 0.2) -> "OptionChain":
        ttms = np.asarray(ttms, dtype=float)
        forwards = np.asarray(forwards, dtype=float)
        if forwards.ndim == 1 and forwards.size != ttms.size and forwards.size > 0 and np.all(forwards == forwards[0]):
            forwards = np.full(ttms.size, forwards[0], dtype=float)
        return cls(ttms=ttms, ids=ids, forwards=forwards, strikes_ttms=[strikes for _ in ttms],
                   bid_ivs=[flat_vol * np.ones_like(strikes) for _ in ttms],
                   ask_ivs=[flat_vol * np.ones_like(strikes) for _ in ttms],
                   optiontypes_ttms=[np.where(strikes >= f, "C", "P") for f in forwards])

    def get_chain_data_as_xy(self) -> Tuple[tuple, List[np.ndarray]]:
        """(ttms, forwards, discfactors, strikes, types), mid vols: the calibration inputs (reference :318-325)"""
        if self.bid_ivs is None or self.ask_ivs is None:
            raise ValueError("calibration needs bid_ivs and ask_ivs on the chain")
        return (self.ttms, self.forwards, self.discfactors, self.strikes_ttms, self.optiontypes_ttms), self.get_mid_vols()

    def get_chain_vegas(self, is_unit_ttm_vega: bool = False) -> List[np.ndarray]:
        """Black vegas at the mid vols, the calibration weights of Eq. (6.3) (reference :263-279; the reference
        delegates to the third-party compute_bsm_vegas_ttms -- forward vega F*pdf(d1)*sqrt(T) here; the objective
        normalises them per slice, so a constant factor is immaterial).  is_unit_ttm_vega evaluates them at T = 1."""
        ttms = np.ones_like(self.ttms) if is_unit_ttm_vega else self.ttms
        return [black_vega(float(f), np.asarray(k, dtype=float), float(t), np.asarray(v, dtype=float))
                for t, f, k, v in zip(ttms, self.forwards, self.strikes_ttms, self.get_mid_vols())]

    def get_chain_atm_vols(self) -> np.ndarray:
        """mid vol of each slice linearly interpolated to the forward (reference :281-286)"""
        return np.array([np.interp(x=f, xp=k, fp=v)
                         for f, k, v in zip(self.forwards, self.strikes_ttms, self.get_mid_vols())])

    def get_slice_varswap_strikes(self, floor_with_atm_vols: bool = True) -> pd.Series:
        """variance-swap strike (as a volatility) per maturity, replicated from the chain's own strikes priced at the
        mid vols (reference :402-426; its Black prices come from the third-party package, here from black_price --
        UNDISCOUNTED, as the replication wants them); floored with the ATM vol by default because a sparse strike
        grid biases the replication low."""
        from ..utils.var_swap_pricer import compute_var_swap_strike
        out = np.zeros_like(self.ttms, dtype=float)
        for idx, (ttm, forward, strikes, types, vols) in enumerate(zip(self.ttms, self.forwards, self.strikes_ttms,
                                                                       self.optiontypes_ttms, self.get_mid_vols())):
            strikes = np.asarray(strikes, dtype=float)
            is_put = np.asarray(types).astype(str) == "P"
            mid = black_price(float(forward), strikes, float(ttm), np.asarray(vols, dtype=float), ~is_put)
            out[idx] = compute_var_swap_strike(puts=pd.Series(mid[is_put], index=strikes[is_put]),
                                               calls=pd.Series(mid[~is_put], index=strikes[~is_put]),
                                               forward=float(forward), ttm=float(ttm))
        if floor_with_atm_vols:
            out = np.maximum(self.get_chain_atm_vols(), out)
        return pd.Series(out, index=self.ttms)

    def compute_model_ivols_from_chain_data(self, model_prices, forwards=None) -> List[np.ndarray]:
        """model prices -> Black implied vols, slice by slice (reference data/option_chain.py:327-346).

        The reference delegates to the third-party `vanilla_option_pricers.infer_bsm_ivols_from_model_chain_prices`,
        which is not available here: this is a textbook Black-76 inversion -- 'C' / 'P', and 'IC' / 'IP' as the vanilla
        inversion of price x forward (an inverse option's Black value is the vanilla value over the forward) -- done by
        libsvmc's host routine svmc_black_implied_vols (safeguarded Newton; the MC calibration loop gets the same inversion from the
        last kernel of its replayed graph instead, logsv_mc_chain_pricer_fixed_randoms(return_ivols=True)).  Parity with the third-party routine is UNPINNED; prices outside the band
        attainable for vols in [1e-6, 10] give NaN."""
        forwards = self.forwards if forwards is None else forwards
        return [black_ivols_native(np.asarray(p, dtype=float), float(t), float(f), np.asarray(k, dtype=float), ty, float(d))
                for p, t, f, k, ty, d in zip(model_prices, self.ttms, forwards, self.strikes_ttms,
                                             self.optiontypes_ttms, self.discfactors)]


def black_ivols_native(prices: np.ndarray, ttm: float, forward: float, strikes: np.ndarray, optiontypes,
                       discfactor: float = 1.0, lo: float = 1e-6, hi: float = 10.0) -> np.ndarray:
    """svmc_black_implied_vols through ctypes; same contract as infer_black_ivols below (the NumPy bisection kept as
    the independent check)"""
    import ctypes as C
    from .. import _lib
    from ..engine import option_type_codes
    codes = option_type_codes(np.asarray(optiontypes).astype(str))       # C, P, IC, IP -> 0..3; ValueError otherwise
    prices = np.ascontiguousarray(prices, dtype=np.float64)
    strikes = np.ascontiguousarray(strikes, dtype=np.float64)
    out = np.empty(strikes.shape, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    _lib.check(_lib.load().svmc_black_implied_vols(prices.ctypes.data_as(dp), strikes.ctypes.data_as(dp),
                                                   codes.ctypes.data_as(C.POINTER(C.c_int8)), strikes.size,
                                                   forward, ttm, discfactor, lo, hi, out.ctypes.data_as(dp)))
    return out


def black_price(forward: float, strikes: np.ndarray, ttm: float, vol: np.ndarray, is_call: np.ndarray,
                discfactor: float = 1.0) -> np.ndarray:
    """Black-76 price of calls / puts on a forward"""
    from scipy.special import ndtr
    sv = np.maximum(vol, 1e-300) * np.sqrt(ttm)
    d1 = np.log(forward / strikes) / sv + 0.5 * sv
    d2 = d1 - sv
    call = discfactor * (forward * ndtr(d1) - strikes * ndtr(d2))
    return np.where(is_call, call, call - discfactor * (forward - strikes))


def black_vega(forward: float, strikes: np.ndarray, ttm: float, vol: np.ndarray) -> np.ndarray:
    """dPrice/dvol of an undiscounted Black-76 option"""
    sv = np.maximum(vol, 1e-300) * np.sqrt(ttm)
    d1 = np.log(forward / strikes) / sv + 0.5 * sv
    return forward * np.exp(-0.5 * d1 * d1) / np.sqrt(2.0 * np.pi) * np.sqrt(ttm)


def infer_black_ivols(prices: np.ndarray, ttm: float, forward: float, strikes: np.ndarray, optiontypes,
                      discfactor: float = 1.0, lo: float = 1e-6, hi: float = 10.0) -> np.ndarray:
    """Black-76 implied vols by bisection on [lo, hi] (monotone in vol; 60 halvings reach 1e-17 of the bracket)"""
    types = np.asarray(optiontypes).astype(str)
    if not np.all(np.isin(types, ("C", "P", "IC", "IP"))):
        raise ValueError("unknown option payoff code")
    is_call = (types == "C") | (types == "IC")
    prices = np.where((types == "IC") | (types == "IP"), np.asarray(prices, dtype=float) * forward, prices)   # inverse: x forward
    a = np.full(strikes.shape, lo)
    b = np.full(strikes.shape, hi)
    ok = (prices > black_price(forward, strikes, ttm, a, is_call, discfactor)) & \
         (prices < black_price(forward, strikes, ttm, b, is_call, discfactor))
    for _ in range(60):
        mid = 0.5 * (a + b)
        below = black_price(forward, strikes, ttm, mid, is_call, discfactor) < prices
        a = np.where(below, mid, a)
        b = np.where(below, b, mid)
    return np.where(ok, 0.5 * (a + b), np.nan)
