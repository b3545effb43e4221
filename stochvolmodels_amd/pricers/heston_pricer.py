"""
Heston Monte Carlo on MI355X: drop-in for the MC part of the reference's pricers/heston_pricer.py
(HestonParams :27-41, model_mc_price_chain :68-87, simulate_terminal_values :89-108,
heston_mc_chain_pricer :285-331, simulate_heston_x_vol_terminal :334-381).

The reference's only scheme is an Euler step with the variance floored at 1e-4; it is the default here
(scheme="euler").  scheme="qe" selects Andersen's QE-M (new capability; validated against the reference's
analytic Heston prices).  `nb_steps_per_year` is exposed on the chain driver with the reference's fixed
value 360 as default; `seed=` and `comm=` as in logsv_pricer.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import dist as svdist
from ..data.option_chain import OptionChain
from ..engine import HESTON_EULER_FLOOR, HESTON_QE, get_engine, marshalled_chain, option_type_codes
from ..mc_chain import price_chain_on_engine, variable_type_code
from ..utils.calibration import ImpliedVolObjective, chain_calibration_weights, minimize_slsqp
from ..utils.config import VariableType
from ..utils.funcs import next_rng_call, set_time_grid, time_grid_steps, timer
from ..analytic import AnalyticGrid, qvar_prices_from_sums, vanilla_prices_from_capped
from ..utils import mgf_pricer as mgfp
from .logsv_pricer import _broadcast_state
from .model_pricer import ModelParams, ModelPricer


@dataclass
class HestonParams(ModelParams):
    """dv = kappa (theta - v) dt + volvol sqrt(v) dW, rho = corr(dS, dv)."""
    v0: float = 0.04
    theta: float = 0.04
    kappa: float = 4.0
    rho: float = -0.5
    volvol: float = 0.4


BTC_HESTON_PARAMS = HestonParams(v0=0.8, theta=1.0, kappa=2.0, rho=0.0, volvol=2.0)


def _scheme_code(scheme) -> int:
    if scheme in (HESTON_EULER_FLOOR, "euler", "euler_floor", None):
        return HESTON_EULER_FLOOR
    if scheme in (HESTON_QE, "qe", "QE"):
        return HESTON_QE
    raise ValueError(f"unknown Heston scheme {scheme!r} (use 'euler' or 'qe')")


# heston_mc_chain_pricer steps all expiries of a multi-expiry chain in one launch (svmc_heston_chain_rng) when True
WHOLE_CHAIN_STEPPING = True
# single-GPU chains go through svmc_heston_chain_price on the engine's own state (one C-ABI call per chain) when True
FUSED_MC_CHAIN_DRIVER = True


class HestonPricer(ModelPricer):

    def price_chain(self, option_chain: OptionChain, params: HestonParams, **kwargs) -> List[np.ndarray]:
        """analytic chain prices by Fourier inversion of the closed-form MGF (reference :52-66)"""
        return heston_chain_pricer(v0=params.v0, theta=params.theta, kappa=params.kappa, volvol=params.volvol,
                                   rho=params.rho, ttms=option_chain.ttms, forwards=option_chain.forwards,
                                   discfactors=option_chain.discfactors, strikes_ttms=option_chain.strikes_ttms,
                                   optiontypes_ttms=option_chain.optiontypes_ttms,
                                   variable_type=kwargs.get("variable_type", VariableType.LOG_RETURN))

    def model_mc_price_chain(self, option_chain: OptionChain, params: HestonParams, nb_path: int = 100000,
                             variable_type: VariableType = VariableType.LOG_RETURN, **kwargs
                             ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        return heston_mc_chain_pricer(v0=params.v0, theta=params.theta, kappa=params.kappa, rho=params.rho,
                                      volvol=params.volvol, ttms=option_chain.ttms, forwards=option_chain.forwards,
                                      discfactors=option_chain.discfactors, strikes_ttms=option_chain.strikes_ttms,
                                      optiontypes_ttms=option_chain.optiontypes_ttms, nb_path=nb_path,
                                      variable_type=variable_type, scheme=kwargs.get("scheme", "euler"),
                                      nb_steps_per_year=kwargs.get("nb_steps_per_year", 360),
                                      seed=kwargs.get("seed"), comm=kwargs.get("comm"), devices=kwargs.get("devices"),
                                      reduce=kwargs.get("reduce"))

    @timer
    def calibrate_model_params_to_chain(self, option_chain: OptionChain, params0: HestonParams = None,
                                        is_vega_weighted: bool = True, is_unit_ttm_vega: bool = False, **kwargs
                                        ) -> HestonParams:
        """fit (v0, theta, kappa, rho, volvol) to the chain's mid vols with the analytic pricer under the Feller
        constraint 2 kappa theta >= volvol^2 (reference :110-181; same start point, bounds and SLSQP options)"""
        p0 = np.array([0.1, 0.1, 2.0, -0.2, 1.0]) if params0 is None else \
            np.array([params0.v0, params0.theta, params0.kappa, params0.rho, params0.volvol])
        bounds = ((0.01, 2.0), (0.01, 2.0), (0.1, 30.0), (-0.99, 0.99), (0.1, 5.0))
        _, market_vols_ttms = option_chain.get_chain_data_as_xy()
        market_vols = np.concatenate(market_vols_ttms).ravel()
        weights = chain_calibration_weights(option_chain, market_vols, is_vega_weighted, is_unit_ttm_vega)

        def parse(pars: np.ndarray) -> HestonParams:
            return HestonParams(v0=pars[0], theta=pars[1], kappa=pars[2], rho=pars[3], volvol=pars[4])

        objective = ImpliedVolObjective(
            lambda pars: self.compute_model_ivols_for_chain(option_chain=option_chain, params=parse(pars)),
            market_vols, weights)
        feller = {"type": "ineq", "fun": lambda pars: 2.0 * pars[2] * pars[1] - pars[4] * pars[4]}
        fit = minimize_slsqp(objective, p0, bounds, feller, disp=bool(kwargs.get("disp", True)))
        self.last_calibration = dict(n_eval=objective.n_eval, objective=objective(fit))
        return parse(fit)

    @timer
    def simulate_terminal_values(self, params: HestonParams, ttm: float = 1.0, nb_path: int = 100000,
                                 x0: float = 0.0, **kwargs) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """returns (x, VARIANCE, qvar) like the reference (:89-108); the x0 argument is ignored there too."""
        nb_steps, dt = time_grid_steps(ttm=ttm, nb_steps_per_year=360)
        rng_seed, call_id = next_rng_call(kwargs.get("seed"))
        eng = get_engine(nb_path)
        eng.fill_state(0.0, params.v0, 0.0)           # constant initial state written on the device (reference :98-107)
        eng.heston_rng(nb_steps, dt, params.theta, params.kappa, params.rho, params.volvol,
                       _scheme_code(kwargs.get("scheme", "euler")), rng_seed, call_id, 0)
        return eng.get_state()


def compute_heston_mgf_grid(v0: float, theta: float, kappa: float, volvol: float, rho: float, ttm: float,
                            phi_grid: np.ndarray, psi_grid: np.ndarray, a_t0: np.ndarray = None, b_t0: np.ndarray = None
                            ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """closed-form Heston log-MGF over the grid, (log_mgf, a_t1, b_t1) (reference :183-214).  A missing a_t0 / b_t0
    is the zero vector: the reference's `None` branches are exactly the formulas at a_t0 = b_t0 = 0."""
    from .. import _lib
    grid = AnalyticGrid(np.asarray(phi_grid), np.asarray(psi_grid), 1)
    try:
        if a_t0 is not None:
            grid.set_a(np.asarray(a_t0, dtype=np.complex128).reshape(-1, 1))
        if b_t0 is not None:
            b0 = np.ascontiguousarray(b_t0, dtype=np.complex128)
            _lib.check(grid.lib.svmc_memcpy_h2d(grid.b.ptr, b0.ctypes.data, b0.nbytes, None))
            _lib.check(grid.lib.svmc_stream_synchronize(None))
        grid.heston_advance(ttm, v0, theta, kappa, volvol, rho, True)
        return grid.get_log_mgf(), grid.get_a().ravel(), grid._down(grid.b, (grid.n,))
    finally:
        grid.close()


def heston_chain_pricer(v0: float, theta: float, kappa: float, volvol: float, rho: float, ttms: np.ndarray,
                        forwards: np.ndarray, strikes_ttms: Sequence[np.ndarray], optiontypes_ttms: Sequence[np.ndarray],
                        discfactors: np.ndarray, variable_type: VariableType = VariableType.LOG_RETURN,
                        vol_scaler: float = None) -> List[np.ndarray]:
    """analytic Heston chain prices (reference :217-282): LOG_RETURN on the 1000-point phi grid, Q_VAR (calls on the
    annualised quadratic variance) on the 40 000-point psi grid"""
    vt = int(getattr(variable_type, "value", variable_type))
    if vt not in (1, 2):
        raise NotImplementedError(f"variable_type={variable_type}")
    if vol_scaler is None:
        vol_scaler = np.minimum(0.3, np.sqrt(v0 * ttms[0]))
    phi_grid, psi_grid, _ = mgfp.get_transform_var_grid(variable_type=variable_type, vol_scaler=vol_scaler)
    grid = AnalyticGrid.acquire(phi_grid, psi_grid, 1)
    try:
        # the chain's launches queued back to back, one download of every expiry's sums at the end (AnalyticGrid.queue_*)
        ks = [int(np.asarray(k).size) for k in strikes_ttms]
        offs = np.concatenate([[0], np.cumsum(ks)]).astype(int)
        grid.reserve_results(int(offs[-1]))
        ttm0 = 0.0
        for i, (ttm, forward, strikes) in enumerate(zip(ttms, forwards, strikes_ttms)):
            grid.heston_advance(ttm - ttm0, v0, theta, kappa, volvol, rho, True)     # zero a, b == the None branch
            if vt == 1:
                grid.queue_capped_sums(float(forward), np.asarray(strikes, dtype=np.float64), int(offs[i]))
            else:
                grid.queue_qvar_sums(float(ttm), np.asarray(strikes, dtype=np.float64), int(offs[i]))
            ttm0 = ttm
        sums = grid.download_results(int(offs[-1]))
        prices = []
        for i, (ttm, forward, discfactor, strikes, types) in enumerate(zip(ttms, forwards, discfactors, strikes_ttms,
                                                                           optiontypes_ttms)):
            if vt == 1:
                prices.append(vanilla_prices_from_capped(sums[offs[i]:offs[i + 1]], float(forward), strikes, types,
                                                         float(discfactor), True))
            else:
                prices.append(qvar_prices_from_sums(sums[offs[i]:offs[i + 1]], float(ttm), types, float(discfactor)))
        return prices
    finally:
        grid.release()


def simulate_heston_x_vol_terminal(ttm: float, x0: np.ndarray, var0: np.ndarray, qvar0: np.ndarray, theta: float,
                                   kappa: float, rho: float, volvol: float, nb_path: int = 100000,
                                   nb_steps_per_year: int = 360, scheme="euler", seed: Optional[int] = None,
                                   W0: Optional[np.ndarray] = None, W1: Optional[np.ndarray] = None,
                                   dt: Optional[float] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """terminal (x, variance, qvar) (reference :334-381).  W0/W1/dt are an extension: supplied UNSCALED normals
    for the Euler scheme, the fixed-randoms route the reference offers only for LogSV."""
    x0, var0, qvar0 = _broadcast_state(x0, var0, qvar0, nb_path)
    code = _scheme_code(scheme)
    eng = get_engine(nb_path)
    eng.set_state(x0, var0, qvar0)
    if W0 is None and W1 is None:
        nb_steps, dt = time_grid_steps(ttm=ttm, nb_steps_per_year=nb_steps_per_year)
        rng_seed, call_id = next_rng_call(seed)
        eng.heston_rng(nb_steps, dt, theta, kappa, rho, volvol, code, rng_seed, call_id, 0)
    else:
        if code != HESTON_EULER_FLOOR:
            raise ValueError("supplied W0/W1 drive the Euler scheme only")
        W0, W1 = np.asarray(W0), np.asarray(W1)
        if W0.shape != W1.shape or W0.ndim != 2 or W0.shape[1] != nb_path or dt is None:
            raise ValueError("W0 and W1 must both have shape [nb_steps, nb_path] and come with dt")
        w0, w1 = eng.upload_randoms((W0, W1))
        eng.heston_w(W0.shape[0], dt, theta, kappa, rho, volvol, w0, w1)
    return eng.get_state()


def heston_mc_chain_pricer(ttms: np.ndarray, forwards: np.ndarray, discfactors: np.ndarray,
                           strikes_ttms: Sequence[np.ndarray], optiontypes_ttms: Sequence[np.ndarray], v0: float,
                           theta: float, kappa: float, rho: float, volvol: float, nb_path: int = 100000,
                           variable_type: VariableType = VariableType.LOG_RETURN, scheme="euler",
                           nb_steps_per_year: int = 360, seed: Optional[int] = None, comm=None, devices=None,
                           reduce: Optional[str] = None) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """chain MC (reference :285-331): state carried slice to slice, 360 steps/yr unless overridden.
    devices=N | [ids]: the paths sharded over that many GPUs of this process (stochvolmodels_amd.multi), as for
    logsv_mc_chain_pricer."""
    vt_code = variable_type_code(variable_type)
    code = _scheme_code(scheme)
    if devices is not None:
        if comm is not None:
            raise ValueError("devices= (one process, several GPUs) and comm= (one process per GPU) are exclusive")
        from ..multi import get_multi_session
        rng_seed, call_id = next_rng_call(seed)
        ms = get_multi_session(devices, nb_path, len(ttms), sum(int(np.asarray(k).size) for k in strikes_ttms), reduce=reduce)
        return ms.price_heston_chain(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, v0, theta, kappa, rho, volvol,
                                     code, nb_steps_per_year, vt_code, rng_seed, call_id)
    comm = comm or svdist.get_default_comm()
    offset, n_local = svdist.shard_range(nb_path, comm.rank, comm.world)
    eng = get_engine(n_local, path_offset=offset)
    rng_seed, call_id = next_rng_call(seed)
    if FUSED_MC_CHAIN_DRIVER and WHOLE_CHAIN_STEPPING and comm.world == 1 and hasattr(eng, "price_heston_chain_fused"):
        # one GPU: ONE call of the fused C driver on the engine's own state (see logsv_mc_chain_pricer)
        ch = marshalled_chain(ttms, forwards, discfactors, strikes_ttms, [option_type_codes(t) for t in optiontypes_ttms])
        prices, stderrs = eng.price_heston_chain_fused(ch, v0, theta, kappa, rho, volvol, code, nb_steps_per_year, vt_code,
                                                       rng_seed, call_id)
        return ([a.reshape(np.shape(k)) for a, k in zip(prices, strikes_ttms)],
                [a.reshape(np.shape(k)) for a, k in zip(stderrs, strikes_ttms)])
    grids, t0 = [], 0.0
    for ttm in ttms:
        nb, dt = time_grid_steps(ttm=ttm - t0, nb_steps_per_year=nb_steps_per_year)
        grids.append((nb, dt))
        t0 = ttm
    step0 = np.concatenate([[0], np.cumsum([g[0] for g in grids])])
    start = (0.0, v0, 0.0)     # every path's start state (reference :303-305) goes to the first stepping launch as constants

    def advance(i: int, forward: float, snap_row: int, qvar_row, spot_ptr: int) -> None:
        nb, dt = grids[i]
        eng.heston_slice_rng(nb, dt, theta, kappa, rho, volvol, code, rng_seed, call_id, int(step0[i]), forward,
                             snap_row, qvar_row, spot_ptr, start=start if i == 0 else None)

    def advance_chain(need_qvar: bool, spot_ptr: int) -> None:
        eng.heston_chain_rng([g[0] for g in grids], [g[1] for g in grids], forwards, theta, kappa, rho, volvol, code,
                             rng_seed, call_id, 0, need_qvar, spot_ptr, start=start)

    return price_chain_on_engine(eng, comm, nb_path, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms,
                                 variable_type, advance,
                                 advance_chain=advance_chain if (WHOLE_CHAIN_STEPPING and len(grids) > 1) else None)
