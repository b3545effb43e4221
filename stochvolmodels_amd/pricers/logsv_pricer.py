"""
LogSV Monte Carlo on MI355X: drop-in for the MC part of the reference's pricers/logsv_pricer.py
(model_mc_price_chain :368-427, simulate_terminal_values :589-611, logsv_mc_chain_pricer :806-867,
simulate_logsv_x_vol_terminal :950-1047, get_randoms_for_chain_valuation :1051-1074,
logsv_mc_chain_pricer_fixed_randoms :1100-1162).

Same names, keyword signatures, defaults and return structure; the arithmetic runs in libsvmc's HIP
kernels (one lane per path, fp64, state resident in HBM across expiries).  Two additions, both optional
keywords: `seed=` (the reference exposes none; default = process seed / set_seed + fresh call id) and
`comm=` (a stochvolmodels_amd.dist communicator to shard paths over GPUs; default = the process default).
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from .. import dist as svdist
from ..data.option_chain import OptionChain
from ..engine import DeviceRandoms, get_engine, marshalled_chain, option_type_codes, payoff_finalize
from ..mc_chain import price_chain_on_engine, variable_type_code
from ..utils.calibration import ImpliedVolObjective, chain_calibration_weights, minimize_slsqp
from ..utils.config import VariableType
from ..utils.funcs import next_rng_call, set_time_grid, time_grid_steps, timer
from ..analytic import AnalyticGrid, qvar_prices_from_sums, vanilla_prices_from_capped
from ..utils import mgf_pricer as mgfp
from .logsv.affine_expansion import ExpansionOrder, _order_code, note_integrator_flags
from .logsv.logsv_params import LogSvParams
from .logsv.vol_moments_ode import fit_model_vol_backbone_to_varswaps
from .model_pricer import ModelPricer

class LogsvModelCalibrationType(Enum):
    """which parameters the calibration solves for (reference :56-68)"""
    PARAMS4 = 1                     # sigma0, theta, beta, volvol; kappa1, kappa2 held at params0
    PARAMS5 = 2                     # sigma0, theta, kappa1, beta, volvol; kappa2 = kappa1 / theta
    PARAMS6 = 3
    PARAMS_WITH_VARSWAP_FIT = 4     # beta, volvol + a vol backbone fitted to variance swaps


class ConstraintsType(Enum):
    """martingale / moment constraints of Theorem 3.7 (reference :70-88)"""
    UNCONSTRAINT = 1
    MMA_MARTINGALE = 2              # kappa2 >= beta
    INVERSE_MARTINGALE = 3          # kappa2 >= 2 beta
    MMA_MARTINGALE_MOMENT4 = 4      # ... and kappa1 + kappa2 theta >= 1.5 (beta^2 + volvol^2)
    INVERSE_MARTINGALE_MOMENT4 = 5


class CalibrationEngine(Enum):
    """where the objective's model vols come from (reference :90-101)"""
    ANALYTIC = 1
    MC = 2
    ROUGH_MC = 3


_FREE_PARAMS = {LogsvModelCalibrationType.PARAMS4: ("sigma0", "theta", "beta", "volvol"),
                LogsvModelCalibrationType.PARAMS5: ("sigma0", "theta", "kappa1", "beta", "volvol"),
                LogsvModelCalibrationType.PARAMS_WITH_VARSWAP_FIT: ("beta", "volvol")}


def _calibration_parser(calibration_type: LogsvModelCalibrationType, params0: LogSvParams,
                        varswap_strikes: Optional[pd.Series] = None):
    """optimizer vector -> LogSvParams.  PARAMS4 keeps params0's kappas, PARAMS5 ties kappa2 = kappa1 / theta
    (LogSvParams(kappa2=None)); PARAMS_WITH_VARSWAP_FIT solves for (beta, volvol) only and, for every candidate,
    refits the vol backbone so that the model reproduces the term structure of `varswap_strikes`
    (vol_moments_ode.fit_model_vol_backbone_to_varswaps); H / nodes / weights always come from params0 (reference
    codec :106-160).  PARAMS6 raises as in the reference."""
    names = _FREE_PARAMS.get(calibration_type)
    if names is None:
        raise NotImplementedError(f"{calibration_type}")
    tied = calibration_type == LogsvModelCalibrationType.PARAMS5
    with_varswaps = calibration_type == LogsvModelCalibrationType.PARAMS_WITH_VARSWAP_FIT

    def parse(pars: np.ndarray) -> LogSvParams:
        fields = dict(sigma0=params0.sigma0, theta=params0.theta, kappa1=params0.kappa1,
                      kappa2=None if tied else params0.kappa2, H=params0.H, nodes=params0.nodes, weights=params0.weights)
        fields.update(zip(names, pars))
        params = LogSvParams(**fields)
        if with_varswaps:
            params.set_vol_backbone(fit_model_vol_backbone_to_varswaps(log_sv_params=params,
                                                                       varswap_strikes=varswap_strikes))
        return params
    return names, parse


def _calibration_constraints(parse, constraints_type: ConstraintsType):
    """SLSQP inequality constraints g(pars) >= 0 (reference :297-332)"""
    def mma(pars):
        p = parse(pars)
        return p.kappa2 - p.beta

    def inverse(pars):
        p = parse(pars)
        return p.kappa2 - 2.0 * p.beta

    def moment4(pars):
        p = parse(pars)
        return p.kappa1 + p.kappa2 * p.theta - 1.5 * (p.beta * p.beta + p.volvol * p.volvol)

    ineq = lambda f: {"type": "ineq", "fun": f}    # noqa: E731
    table = {ConstraintsType.UNCONSTRAINT: None,
             ConstraintsType.MMA_MARTINGALE: ineq(mma),
             ConstraintsType.INVERSE_MARTINGALE: ineq(inverse),
             ConstraintsType.MMA_MARTINGALE_MOMENT4: (ineq(mma), ineq(moment4)),
             ConstraintsType.INVERSE_MARTINGALE_MOMENT4: (ineq(inverse), ineq(moment4))}
    if constraints_type not in table:
        raise NotImplementedError(f"{constraints_type}")
    return table[constraints_type]


# logsv_mc_chain_pricer steps all expiries of a multi-expiry chain in one launch (svmc_logsv_chain_rng) when True,
# slice by slice otherwise (same bits)
WHOLE_CHAIN_STEPPING = True
# single-GPU chains on resident randoms go through svmc_logsv_chain_price_fixed (one C++ call per chain) when True
FUSED_FIXED_RANDOMS_DRIVER = True
# single-GPU chains with on-device randoms go through svmc_logsv_chain_price on the engine's own state (one C-ABI call per chain)
# when True; False (or WHOLE_CHAIN_STEPPING off, or several ranks): the phase-by-phase Python driver mc_chain.price_chain_on_engine
FUSED_MC_CHAIN_DRIVER = True
# rough LogSV chains: every expiry in one stepping launch (svmc_rough_logsv_chain); False = one launch per expiry (A/B, tests)
ROUGH_CHAIN_ONE_LAUNCH = True

LOGSV_BTC_PARAMS = LogSvParams(sigma0=0.8376, theta=1.0413, kappa1=3.1844, kappa2=3.058, beta=0.1514, volvol=1.8458)


# Grid points the coefficient-ODE integrator gave up on in the LAST analytic chain pricing of this process (step floor / try
# cap, csrc/svmc_analytic.hip): an int for logsv_chain_pricer, one count per parameter set for the batched pricer.  Zero for
# every sane parameter vector; non-zero means the inversion dropped those terms (as the reference's np.nansum drops NaN terms)
# and the prices are bounded but NOT the model's -- a RuntimeWarning says so, and a calibrator can test this to penalise the
# evaluation (the reference's solve_ivp returns whatever state it reached, without a signal).
LAST_ANALYTIC_GIVEN_UP = 0


def _note_given_up(count, n_grid: int) -> None:
    global LAST_ANALYTIC_GIVEN_UP
    LAST_ANALYTIC_GIVEN_UP = count
    if np.any(np.asarray(count) > 0):
        import warnings
        warnings.warn(f"analytic LogSV pricer: the coefficient ODEs were given up on {np.asarray(count).tolist()} of {n_grid} "
                      f"transform-grid points (stiff beyond the try cap, or blowing up before the expiry): those terms are dropped "
                      f"from the inversion and the prices are not the model's -- pricers.logsv_pricer.LAST_ANALYTIC_GIVEN_UP",
                      RuntimeWarning, stacklevel=3)


def logsv_chain_pricer_batch(params_list: Sequence[LogSvParams], ttms: np.ndarray, forwards: np.ndarray,
                             discfactors: np.ndarray, strikes_ttms: Sequence[np.ndarray],
                             optiontypes_ttms: Sequence[np.ndarray], is_spot_measure: bool = True,
                             expansion_order: ExpansionOrder = ExpansionOrder.SECOND, vol_scaler: float = None, **kwargs
                             ) -> List[List[np.ndarray]]:
    """logsv_chain_pricer (LOG_RETURN, numerical ODE route) for SEVERAL parameter sets on one chain, all sets advanced
    by one launch per expiry and inverted by one launch per expiry: [set][expiry] -> prices.  Each set keeps its own
    transform grid (set_vol_scaler follows its sigma0, reference :664-666); results are bit-identical to one
    logsv_chain_pricer call per set.  Not in the reference API: the batched form of its per-set loop (config C5's five
    sets; the bumped vectors of a finite-difference gradient)."""
    from ..analytic import AnalyticGridBatch
    order = _order_code(expansion_order)
    grids = [mgfp.get_transform_var_grid(variable_type=VariableType.LOG_RETURN, is_spot_measure=is_spot_measure,
                                         vol_scaler=(set_vol_scaler(sigma0=p.sigma0, ttm=np.min(ttms))
                                                     if vol_scaler is None else vol_scaler)) for p in params_list]
    batch = AnalyticGridBatch.acquire([g[0] for g in grids], [g[1] for g in grids], 5 if order == 2 else 3)
    try:
        # every expiry's launches are queued back to back (advance, invert, advance, invert, ...) and the sums come back in ONE
        # download at the end of the chain: no host round trip between expiries
        ks = [int(np.asarray(k).size) for k in strikes_ttms]
        n_sets = len(params_list)
        offs = np.concatenate([[0], np.cumsum([n_sets * k for k in ks])]).astype(int)
        batch.reserve_results(int(offs[-1]))
        ttm0 = 0.0
        for i, (ttm, forward, strikes) in enumerate(zip(ttms, forwards, strikes_ttms)):
            rows = np.array([[p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol, p.get_vol_backbone_eta(tau=ttm), 0.0]
                             for p in params_list])
            batch.logsv_advance(ttm - ttm0, rows, is_spot_measure, order, rtol=kwargs.get("ode_rtol"),
                                atol=kwargs.get("ode_atol"))
            batch.queue_capped_sums(float(forward), np.asarray(strikes, dtype=np.float64), int(offs[i]))
            ttm0 = ttm
        sums = batch.download_results(int(offs[-1]))
        _note_given_up(batch.last_given_up, len(grids[0][0]))
        out = [[] for _ in params_list]
        for i, (forward, strikes, types, discfactor) in enumerate(zip(forwards, strikes_ttms, optiontypes_ttms, discfactors)):
            capped = sums[offs[i]:offs[i + 1]].reshape(n_sets, ks[i])
            for s in range(n_sets):
                out[s].append(vanilla_prices_from_capped(capped[s], float(forward), strikes, types, float(discfactor),
                                                         is_spot_measure))
        return out
    finally:
        batch.release()


class LogSVPricer(ModelPricer):

    def price_chain_batch(self, option_chain: OptionChain, params_list: Sequence[LogSvParams], is_spot_measure: bool = True,
                          **kwargs) -> List[List[np.ndarray]]:
        """price_chain for several parameter sets in one batch of launches (logsv_chain_pricer_batch)"""
        return logsv_chain_pricer_batch(params_list=params_list, ttms=option_chain.ttms, forwards=option_chain.forwards,
                                        discfactors=option_chain.discfactors, strikes_ttms=option_chain.strikes_ttms,
                                        optiontypes_ttms=option_chain.optiontypes_ttms, is_spot_measure=is_spot_measure,
                                        **kwargs)

    def price_chain(self, option_chain: OptionChain, params: LogSvParams, is_spot_measure: bool = True, **kwargs
                    ) -> List[np.ndarray]:
        """analytic chain prices by Fourier inversion of the affine expansion (reference :345-366)"""
        return logsv_chain_pricer(params=params, ttms=option_chain.ttms, forwards=option_chain.forwards,
                                  discfactors=option_chain.discfactors, strikes_ttms=option_chain.strikes_ttms,
                                  optiontypes_ttms=option_chain.optiontypes_ttms, is_spot_measure=is_spot_measure,
                                  **kwargs)

    @timer
    def model_mc_price_chain(self, option_chain: OptionChain, params: LogSvParams, is_spot_measure: bool = True,
                             variable_type: VariableType = VariableType.LOG_RETURN, nb_path: int = 100000,
                             nb_steps: Optional[int] = None, **kwargs
                             ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """price an option chain by Monte Carlo.  `nb_steps` is steps PER YEAR and defaults to
        int(360*max(ttms)) + 1, exactly the reference's rule (:427)."""
        if kwargs.get("use_rough_mc"):
            if "seed" not in kwargs:
                raise AssertionError("use_rough_mc requires seed=")          # reference :391
            if params.nodes is None or params.weights is None:
                raise ValueError("rough MC needs params.nodes / params.weights (LogSvParams.approximate_kernel)")
            common = dict(ttms=option_chain.ttms, forwards=option_chain.forwards, discfactors=option_chain.discfactors,
                          strikes_ttms=option_chain.strikes_ttms, optiontypes_ttms=option_chain.optiontypes_ttms,
                          sigma0=params.sigma0, theta=params.theta, kappa1=params.kappa1, kappa2=params.kappa2,
                          beta=params.beta, orthog_vol=params.volvol, weights=params.weights, nodes=params.nodes,
                          variable_type=variable_type, comm=kwargs.get("comm"),
                          normalize_stderr=bool(kwargs.get("normalize_stderr", False)))
            if kwargs.get("device_rng"):      # addition: no host randoms at all
                return rough_logsv_mc_chain_pricer(nb_path=nb_path, nb_steps_per_year=nb_steps or 360,
                                                   seed=kwargs["seed"], **common)
            Z0, Z1, grid_ttms = get_randoms_for_rough_vol_chain_valuation(ttms=option_chain.ttms, nb_path=nb_path,
                                                                          nb_steps_per_year=nb_steps or 360,
                                                                          seed=kwargs["seed"])
            return rough_logsv_mc_chain_pricer_fixed_randoms(Z0=Z0, Z1=Z1, timegrids=grid_ttms, **common)
        etas = params.get_vol_backbone_etas(ttms=option_chain.ttms)
        return logsv_mc_chain_pricer(v0=params.sigma0, theta=params.theta, kappa1=params.kappa1,
                                     kappa2=params.kappa2, beta=params.beta, volvol=params.volvol,
                                     vol_backbone_etas=etas, ttms=option_chain.ttms, forwards=option_chain.forwards,
                                     discfactors=option_chain.discfactors, strikes_ttms=option_chain.strikes_ttms,
                                     optiontypes_ttms=option_chain.optiontypes_ttms, is_spot_measure=is_spot_measure,
                                     variable_type=variable_type, nb_path=nb_path,
                                     nb_steps_per_year=nb_steps or int(360 * np.max(option_chain.ttms)) + 1,
                                     seed=kwargs.get("seed"), comm=kwargs.get("comm"), devices=kwargs.get("devices"),
                                     reduce=kwargs.get("reduce"))

    def set_vol_scaler(self, option_chain: OptionChain) -> float:
        """transform-grid scaler from the chain's first ATM vol, held fixed over a calibration (reference :429-438)"""
        return set_vol_scaler(sigma0=option_chain.get_chain_atm_vols()[0], ttm=option_chain.ttms[0])

    @timer
    def calibrate_model_params_to_chain(self, option_chain: OptionChain, params0: LogSvParams,
                                        params_min: LogSvParams = LogSvParams(sigma0=0.1, theta=0.1, kappa1=0.25,
                                                                              kappa2=0.25, beta=-3.0, volvol=0.2),
                                        params_max: LogSvParams = LogSvParams(sigma0=1.5, theta=1.5, kappa1=10.0,
                                                                              kappa2=10.0, beta=3.0, volvol=3.0),
                                        is_vega_weighted: bool = True, is_unit_ttm_vega: bool = False,
                                        model_calibration_type: LogsvModelCalibrationType = LogsvModelCalibrationType.PARAMS5,
                                        constraints_type: ConstraintsType = ConstraintsType.UNCONSTRAINT,
                                        calibration_engine: CalibrationEngine = CalibrationEngine.ANALYTIC,
                                        nb_path: int = 100000, nb_steps: int = 360, seed: int = 10, **kwargs
                                        ) -> LogSvParams:
        """fit the model to the chain's mid implied vols: SLSQP on the vega-weighted squared vol error of Eq. (6.3)
        (reference :440-557).  The MC engines draw the reference's RandomState(seed) randoms once, upload this rank's
        columns to HBM once, and every objective evaluation re-prices the chain from those resident randoms -- the
        optimizer loop is kernel-bound, not PCIe-bound.  `kwargs`: comm=, disp= (SLSQP printing, default True as in
        the reference).  `self.last_calibration` keeps {"n_eval", "objective"} of the run."""
        vol_scaler = self.set_vol_scaler(option_chain=option_chain)
        _, market_vols_ttms = option_chain.get_chain_data_as_xy()
        market_vols = np.concatenate(market_vols_ttms).ravel()
        weights = chain_calibration_weights(option_chain, market_vols, is_vega_weighted, is_unit_ttm_vega)
        varswap_strikes = (option_chain.get_slice_varswap_strikes(floor_with_atm_vols=True)
                           if model_calibration_type == LogsvModelCalibrationType.PARAMS_WITH_VARSWAP_FIT else None)
        names, parse = _calibration_parser(model_calibration_type, params0, varswap_strikes)
        p0 = np.array([getattr(params0, n) for n in names], dtype=float)
        bounds = tuple((getattr(params_min, n), getattr(params_max, n)) for n in names)
        comm = kwargs.get("comm")
        chain_args = dict(ttms=option_chain.ttms, forwards=option_chain.forwards, discfactors=option_chain.discfactors,
                          strikes_ttms=option_chain.strikes_ttms, optiontypes_ttms=option_chain.optiontypes_ttms)
        resident = None
        model_vols_batch = None
        if calibration_engine == CalibrationEngine.ANALYTIC:
            # ode_rtol= / ode_atol=: the tolerances of the coefficient ODEs for this fit (default: the pricers' 1e-10 / 1e-12;
            # looser settings buy little since the 8th-order pair: stochvolmodels_amd/analytic.py)
            tol = {k: kwargs[k] for k in ("ode_rtol", "ode_atol") if k in kwargs}

            def model_vols(pars):
                return self.compute_model_ivols_for_chain(option_chain=option_chain, params=parse(pars),
                                                          vol_scaler=vol_scaler, **tol)

            def model_vols_batch(pars_list):
                # the n bumped vectors of SLSQP's forward-difference gradient through ONE batch of launches per expiry
                # (logsv_chain_pricer_batch: the sets advance together; bit-identical to one call per set)
                prices = self.price_chain_batch(option_chain=option_chain, params_list=[parse(p) for p in pars_list],
                                                vol_scaler=vol_scaler, **tol)
                return [option_chain.compute_model_ivols_from_chain_data(model_prices=pr) for pr in prices]
            if not kwargs.get("batched_gradient", True):
                model_vols_batch = None
        elif calibration_engine == CalibrationEngine.MC:
            if kwargs.get("device_randoms", False):
                # the fixed randoms are the counter-based stream of `seed` instead of NumPy's RandomState arrays (about a
                # second of host draws per 10^5 paths, 582 MB of upload at 10^5 x 364): same estimator, another sample.
                # device_randoms=True keeps NOTHING resident -- every objective evaluation regenerates the draws in
                # registers (svmc_logsv_chain_price_frozen_sets: no HBM term in the stepping, the draw shared by the
                # bumped parameter sets of a gradient); device_randoms="hbm" materialises them in HBM as round 4 did
                resident = draw_fixed_randoms_on_device(ttms=option_chain.ttms, nb_path=nb_path, nb_steps_per_year=nb_steps,
                                                        seed=seed, comm=comm, in_hbm=kwargs["device_randoms"] == "hbm")
            else:
                resident = upload_fixed_randoms(*get_randoms_for_chain_valuation(
                    ttms=option_chain.ttms, nb_path=nb_path, nb_steps_per_year=nb_steps, seed=seed), comm=comm)

            def model_vols(pars):
                # prices AND their implied vols in one call: on one GPU the inversion is the last kernel of the replayed
                # graph (svmc_logsv_chain_price_fixed_iv), so an objective evaluation is one launch and one wait
                p = parse(pars)
                _, _, ivols = logsv_mc_chain_pricer_fixed_randoms(
                    W0s=resident, W1s=None, dts=None, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2,
                    beta=p.beta, volvol=p.volvol, vol_backbone_etas=p.get_vol_backbone_etas(ttms=option_chain.ttms),
                    comm=comm, return_ivols=True, **chain_args)
                return ivols

            def model_vols_batch(pars_list):
                # the bumped vectors of SLSQP's forward-difference gradient in one replay: every lane steps all of them on
                # each pair of normals it reads (svmc_logsv_chain_price_fixed_sets) -- bit-identical to one call per vector
                out = logsv_mc_chain_pricer_fixed_randoms_batch(params_list=[parse(p) for p in pars_list], W0s=resident,
                                                                return_ivols=True, comm=comm, **chain_args)
                return [res[2] for res in out]
            if not kwargs.get("batched_gradient", True):
                model_vols_batch = None
        elif calibration_engine == CalibrationEngine.ROUGH_MC:
            if kwargs.get("device_randoms", False):
                # as for the MC engine: Z0 / Z1 drawn in HBM instead of by NumPy (same grids, another sample)
                grids = [set_time_grid(ttm=ttm, nb_steps_per_year=nb_steps)[2] for ttm in option_chain.ttms]
                comm_ = comm or svdist.get_default_comm()
                offset, n_local = svdist.shard_range(nb_path, comm_.rank, comm_.world)
                get_engine(n_local, path_offset=offset)
                resident = DeviceRandoms.drawn_on_device([grids[-1].size - 1], [0.0], nb_path, n_local, offset, seed)
            else:
                Z0, Z1, grids = get_randoms_for_rough_vol_chain_valuation(ttms=option_chain.ttms, nb_path=nb_path,
                                                                          nb_steps_per_year=nb_steps, seed=seed)
                resident = upload_rough_randoms(Z0, Z1, comm=comm)

            def model_vols(pars):
                p = parse(pars)
                prices, _ = rough_logsv_mc_chain_pricer_fixed_randoms(
                    Z0=resident, Z1=None, sigma0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2,
                    beta=p.beta, orthog_vol=p.volvol, weights=p.weights, nodes=p.nodes, timegrids=grids, comm=comm,
                    **chain_args)
                return option_chain.compute_model_ivols_from_chain_data(model_prices=prices)
        else:
            raise NotImplementedError(f"{calibration_engine}")
        objective = ImpliedVolObjective(model_vols, market_vols, weights, model_vols_batch=model_vols_batch, bounds=bounds)
        try:
            fit = minimize_slsqp(objective, p0, bounds, _calibration_constraints(parse, constraints_type),
                                 disp=bool(kwargs.get("disp", True)),
                                 jac=objective.gradient if model_vols_batch is not None else None)
            self.last_calibration = dict(n_eval=objective.n_eval, n_gradient_batches=objective.n_batches,
                                         objective=objective(fit))
        finally:
            if resident is not None:
                resident.free()
        return parse(fit)

    @timer
    def simulate_vol_paths(self, params: LogSvParams, brownians: np.ndarray = None, ttm: float = 1.0,
                           nb_path: int = 100000, is_spot_measure: bool = True, nb_steps: int = None,
                           year_days: int = 360, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        """volatility paths on the time grid (reference :560-587).  As in the reference, `nb_steps` (default
        ceil(year_days*ttm)) is handed on as steps PER YEAR."""
        nb_steps = nb_steps or int(np.ceil(year_days * ttm))
        return simulate_vol_paths(ttm=ttm, v0=params.sigma0, theta=params.theta, kappa1=params.kappa1,
                                  kappa2=params.kappa2, beta=params.beta, volvol=params.volvol, nb_path=nb_path,
                                  is_spot_measure=is_spot_measure, nb_steps_per_year=nb_steps, brownians=brownians,
                                  **kwargs)

    def vol_path_moments(self, params: LogSvParams, ttm: float = 1.0, nb_path: int = 100000, is_spot_measure: bool = True,
                         nb_steps: int = None, year_days: int = 360, n_terms: int = 4, **kwargs) -> dict:
        """per-time-step Monte Carlo moments of the volatility paths, computed on the device (module-level vol_path_moments):
        the reduction the reference's scripts apply to simulate_vol_paths' array, without the 8 bytes per path-step of PCIe"""
        nb_steps = nb_steps or int(np.ceil(year_days * ttm))
        return vol_path_moments(ttm=ttm, v0=params.sigma0, theta=params.theta, kappa1=params.kappa1, kappa2=params.kappa2,
                                beta=params.beta, volvol=params.volvol, is_spot_measure=is_spot_measure, nb_path=nb_path,
                                nb_steps_per_year=nb_steps, n_terms=n_terms, **kwargs)

    @timer
    def simulate_terminal_values(self, params: LogSvParams, ttm: float = 1.0, nb_path: int = 100000,
                                 is_spot_measure: bool = True, **kwargs
                                 ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        # same as simulate_logsv_x_vol_terminal(x0=zeros, sigma0=sigma0*ones, qvar0=zeros, ...) (reference :600-610),
        # with the constant initial state written on the device instead of uploaded
        nb_steps, dt = time_grid_steps(ttm=ttm, nb_steps_per_year=360)
        rng_seed, call_id = next_rng_call(kwargs.get("seed"))
        eng = get_engine(nb_path)
        eng.fill_state(0.0, params.sigma0, 0.0)
        eng.logsv_rng(nb_steps, dt, params.theta, params.kappa1, params.kappa2, params.beta, params.volvol, 1.0,
                      is_spot_measure, rng_seed, call_id, 0)
        return eng.get_state()


def set_vol_scaler(sigma0: float, ttm: float) -> float:
    """transform-grid scaler from the ATM vol and the shortest maturity, floored at two weeks (reference :664-666)"""
    return sigma0 * np.sqrt(np.minimum(np.min(ttm), 0.5 / 12.0))


def logsv_chain_pricer(params: LogSvParams, ttms: np.ndarray, forwards: np.ndarray, discfactors: np.ndarray,
                       strikes_ttms: Sequence[np.ndarray], optiontypes_ttms: Sequence[np.ndarray],
                       is_stiff_solver: bool = False, is_analytic: bool = False, is_spot_measure: bool = True,
                       expansion_order: ExpansionOrder = ExpansionOrder.SECOND,
                       variable_type: VariableType = VariableType.LOG_RETURN, vol_scaler: float = None, **kwargs
                       ) -> List[np.ndarray]:
    """analytic LogSV chain prices (reference :669-739): per expiry one launch integrating the coefficient ODEs of every
    transform-grid point from the previous expiry's A, then one launch of per-strike Simpson sums.  LOG_RETURN uses
    the 1000-point phi grid; Q_VAR (calls on the annualised quadratic variance) the 40 000-point psi grid."""
    vt = int(getattr(variable_type, "value", variable_type))
    if vt not in (1, 2):
        raise NotImplementedError
    note_integrator_flags(is_stiff_solver, is_analytic)       # both accepted: one device integrator answers them (warned once)
    order = _order_code(expansion_order)
    if vol_scaler is None:
        vol_scaler = set_vol_scaler(sigma0=params.sigma0, ttm=np.min(ttms))
    phi_grid, psi_grid, _ = mgfp.get_transform_var_grid(variable_type=variable_type, is_spot_measure=is_spot_measure,
                                                        vol_scaler=vol_scaler)
    grid = AnalyticGrid.acquire(phi_grid, psi_grid, 5 if order == 2 else 3)
    try:
        # the chain's launches queued back to back, one download of every expiry's sums at the end (AnalyticGrid.queue_*)
        ks = [int(np.asarray(k).size) for k in strikes_ttms]
        offs = np.concatenate([[0], np.cumsum(ks)]).astype(int)
        grid.reserve_results(int(offs[-1]))
        ttm0 = 0.0
        for i, (ttm, forward, strikes) in enumerate(zip(ttms, forwards, strikes_ttms)):
            eta = params.get_vol_backbone_eta(tau=ttm)
            grid.logsv_advance(ttm - ttm0, params.sigma0, params.theta, params.kappa1, params.kappa2, params.beta,
                               params.volvol, is_spot_measure, order, eta, rtol=kwargs.get("ode_rtol"),
                               atol=kwargs.get("ode_atol"))
            if vt == 1:
                grid.queue_capped_sums(float(forward), np.asarray(strikes, dtype=np.float64), int(offs[i]))
            else:
                grid.queue_qvar_sums(float(ttm), np.asarray(strikes, dtype=np.float64), int(offs[i]))
            ttm0 = ttm
        sums = grid.download_results(int(offs[-1]))
        _note_given_up(grid.last_given_up, grid.n)
        prices = []
        for i, (ttm, forward, strikes, types, discfactor) in enumerate(zip(ttms, forwards, strikes_ttms, optiontypes_ttms,
                                                                           discfactors)):
            if vt == 1:
                prices.append(vanilla_prices_from_capped(sums[offs[i]:offs[i + 1]], float(forward), strikes, types,
                                                         float(discfactor), is_spot_measure))
            else:
                prices.append(qvar_prices_from_sums(sums[offs[i]:offs[i + 1]], float(ttm), types, float(discfactor)))
        return prices
    finally:
        grid.release()


def _broadcast_state(x0, vol0, qvar0, nb_path):
    """length-1 inputs are initial values (x0 -> x0*zeros, qvar0 -> zeros, vol0 -> vol0*ones); otherwise the
    vectors must have nb_path entries (reference :1007-1020, AssertionError)."""
    x0, vol0, qvar0 = np.asarray(x0, dtype=np.float64), np.asarray(vol0, dtype=np.float64), \
        np.asarray(qvar0, dtype=np.float64)
    if x0.shape[0] == 1:
        x0 = x0 * np.zeros(nb_path)
    else:
        assert x0.shape[0] == nb_path
    if qvar0.shape[0] == 1:
        qvar0 = np.zeros(nb_path)
    else:
        assert qvar0.shape[0] == nb_path
    if vol0.shape[0] == 1:
        vol0 = vol0 * np.ones(nb_path)
    else:
        assert vol0.shape[0] == nb_path
    return x0, vol0, qvar0


def simulate_logsv_x_vol_terminal(ttm: float, x0: np.ndarray, sigma0: np.ndarray, qvar0: np.ndarray, theta: float,
                                  kappa1: float, kappa2: float, beta: float, volvol: float,
                                  vol_backbone_eta: float = 1.0, is_spot_measure: bool = True, nb_path: int = 100000,
                                  nb_steps_per_year: int = 360, W0: Optional[np.ndarray] = None,
                                  W1: Optional[np.ndarray] = None, dt: Optional[float] = None,
                                  seed: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """terminal (x, sigma, qvar) after one slice of Eq. (3.59) (reference :950-1047).  W0/W1 are UNSCALED
    N(0,1) of shape [nb_steps, nb_path] with their `dt`; when omitted the increments are drawn on device."""
    x0, sigma0, qvar0 = _broadcast_state(x0, sigma0, qvar0, nb_path)
    eng = get_engine(nb_path)
    eng.set_state(x0, sigma0, qvar0)
    if W0 is None and W1 is None:
        nb_steps, dt = time_grid_steps(ttm=ttm, nb_steps_per_year=nb_steps_per_year)
        rng_seed, call_id = next_rng_call(seed)
        eng.logsv_rng(nb_steps, dt, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta, is_spot_measure,
                      rng_seed, call_id, 0)
    else:
        W0, W1 = np.asarray(W0), np.asarray(W1)
        if W0.shape != W1.shape or W0.ndim != 2 or W0.shape[1] != nb_path:
            raise ValueError("W0 and W1 must both have shape [nb_steps, nb_path]")
        if dt is None:
            raise ValueError("dt must be supplied with W0 and W1")
        w0, w1 = eng.upload_randoms((W0, W1))
        eng.logsv_w(W0.shape[0], dt, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta, is_spot_measure, w0, w1)
    return eng.get_state()


def simulate_vol_paths(ttm: float, v0: float, theta: float, kappa1: float, kappa2: float, beta: float, volvol: float,
                       is_spot_measure: bool = True, nb_path: int = 100000, nb_steps_per_year: int = 360,
                       brownians: np.ndarray = None, seed: Optional[int] = None, return_device: bool = False,
                       out: Optional[np.ndarray] = None, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
    """sigma_t of shape (nb_steps + 1, nb_path), first row = v0, and the time grid (reference :870-947).
    `brownians` are the reference's SCALED increments sqrt(dt)*N(0,1), shape (nb_steps, nb_path).
    The array is 8 bytes per path-step -- 8.6 GB at 2^20 x 1024, two milliseconds to compute: by default it comes back as
    the reference's NumPy array through a pinned, pipelined download (into `out` when given); return_device=True leaves it in
    HBM as an engine.DeviceArray (zero-copy into torch / cupy, `.row_moments()`, `.numpy()`) -- see vol_path_moments()."""
    nb_steps, dt, grid_t = set_time_grid(ttm=ttm, nb_steps_per_year=nb_steps_per_year)
    if brownians is not None:
        brownians = np.asarray(brownians)
        if brownians.shape != (nb_steps, nb_path):
            raise ValueError("brownians must have shape (nb_steps, nb_path)")
    rng_seed, call_id = next_rng_call(seed)
    eng = get_engine(nb_path)
    sigma_t = eng.logsv_vol_paths(nb_steps, dt, v0, theta, kappa1, kappa2, beta, volvol, is_spot_measure, rng_seed,
                                  call_id, brownians=brownians, return_device=return_device, out_host=out)
    return sigma_t, grid_t


def vol_path_moments(ttm: float, v0: float, theta: float, kappa1: float, kappa2: float, beta: float, volvol: float,
                     is_spot_measure: bool = True, nb_path: int = 100000, nb_steps_per_year: int = 360, n_terms: int = 4,
                     center: Optional[float] = None, with_qvar: bool = False, seed: Optional[int] = None) -> dict:
    """What the reference's users of simulate_vol_paths compute from the array (papers/logsv_model_with_quadratic_drift/
    moments_vol_qvar.py:48, :98-104), without the array ever leaving the GPU: per time step the Monte Carlo mean and
    population standard deviation over the paths of (sigma_t - center)^k, k = 1 .. n_terms (center = theta by default, as in
    plot_vol_moments_vs_mc), and with with_qvar those of the expanding time average of sigma_t^2.  -> {"grid_t", "mean"
    [nb_steps + 1][n_terms], "std" (same shape; divide by sqrt(nb_path) for the standard error), "qvar_mean", "qvar_std"}."""
    sigma_t, grid_t = simulate_vol_paths(ttm=ttm, v0=v0, theta=theta, kappa1=kappa1, kappa2=kappa2, beta=beta, volvol=volvol,
                                         is_spot_measure=is_spot_measure, nb_path=nb_path, nb_steps_per_year=nb_steps_per_year,
                                         seed=seed, return_device="view")      # the engine's cached buffer: no 8.6 GB allocation per call
    try:
        mean, std = sigma_t.row_moments(center=theta if center is None else center, n_moments=n_terms)
        out = {"grid_t": grid_t, "mean": mean, "std": std}
        if with_qvar:
            q = sigma_t.expanding_mean_of_squares()
            try:
                qm, qs = q.row_moments(center=0.0, n_moments=1)
            finally:
                q.free()
            out["qvar_mean"], out["qvar_std"] = qm[:, 0], qs[:, 0]
        return out
    finally:
        sigma_t.free()
        get_engine(nb_path).trim_bulk()        # the views are done: cached path arrays beyond an eighth of the device's memory go back


def logsv_mc_chain_pricer(ttms: np.ndarray, forwards: np.ndarray, discfactors: np.ndarray,
                          strikes_ttms: Sequence[np.ndarray], optiontypes_ttms: Sequence[np.ndarray], v0: float,
                          theta: float, kappa1: float, kappa2: float, beta: float, volvol: float,
                          vol_backbone_etas: np.ndarray, is_spot_measure: bool = True, nb_path: int = 100000,
                          nb_steps_per_year: int = 360, variable_type: VariableType = VariableType.LOG_RETURN,
                          seed: Optional[int] = None, comm=None, devices=None, reduce: Optional[str] = None
                          ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """chain MC with on-device randoms (reference :806-867): per slice nb_steps_i = int((T_i - T_{i-1})*spy) + 1.
    devices=N | [ids]: shard the paths over that many GPUs of THIS process (stochvolmodels_amd.multi: one session and one
    host thread per device inside libsvmc; reduce='auto' | 'host' | 'rccl' picks the all-reduce transport) -- the same
    numbers as one device up to the order of the final additions."""
    vt_code = variable_type_code(variable_type)
    if devices is not None:
        if comm is not None:
            raise ValueError("devices= (one process, several GPUs) and comm= (one process per GPU) are exclusive")
        from ..multi import get_multi_session
        rng_seed, call_id = next_rng_call(seed)
        ms = get_multi_session(devices, nb_path, len(ttms), sum(int(np.asarray(k).size) for k in strikes_ttms), reduce=reduce)
        return ms.price_logsv_chain(ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms, v0, theta, kappa1, kappa2,
                                    beta, volvol, vol_backbone_etas, is_spot_measure, nb_steps_per_year, vt_code, rng_seed,
                                    call_id)
    comm = comm or svdist.get_default_comm()
    rng_seed, call_id = next_rng_call(seed)
    if FUSED_MC_CHAIN_DRIVER and WHOLE_CHAIN_STEPPING and comm.world == 1:
        eng = get_engine(nb_path)
        if hasattr(eng, "price_logsv_chain_fused"):
            # one GPU: the whole chain is ONE call of the fused C driver on the engine's own state (the kernels of the route
            # below in the same order: the same bits; -21 us of interpreter and ctypes per call)
            ch = marshalled_chain(ttms, forwards, discfactors, strikes_ttms, [option_type_codes(t) for t in optiontypes_ttms])
            prices, stderrs = eng.price_logsv_chain_fused(ch, v0, theta, kappa1, kappa2, beta, volvol, vol_backbone_etas,
                                                          is_spot_measure, nb_steps_per_year, vt_code, rng_seed, call_id)
            return ([_shaped_like(a, k) for a, k in zip(prices, strikes_ttms)],
                    [_shaped_like(a, k) for a, k in zip(stderrs, strikes_ttms)])
    grids, t0 = [], 0.0
    for ttm in ttms:
        nb, dt = time_grid_steps(ttm=ttm - t0, nb_steps_per_year=nb_steps_per_year)
        grids.append((nb, dt))
        t0 = ttm
    return _logsv_mc_chain_on_grids(grids, rng_seed, call_id, comm, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms,
                                    v0, theta, kappa1, kappa2, beta, volvol, vol_backbone_etas, is_spot_measure, nb_path,
                                    variable_type)


def _logsv_mc_chain_on_grids(grids, rng_seed: int, call_id: int, comm, ttms, forwards, discfactors, strikes_ttms,
                             optiontypes_ttms, v0, theta, kappa1, kappa2, beta, volvol, vol_backbone_etas, is_spot_measure,
                             nb_path, variable_type) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """the on-device-RNG chain on explicit per-expiry grids [(nb_steps_i, dt_i)] and an explicit stream (seed, call id):
    logsv_mc_chain_pricer after its bookkeeping, and what a chain on FROZEN randoms (DeviceRandoms.frozen) runs when it is
    sharded over ranks"""
    offset, n_local = svdist.shard_range(nb_path, comm.rank, comm.world)
    eng = get_engine(n_local, path_offset=offset)
    step0 = [0]
    for g in grids:
        step0.append(step0[-1] + g[0])
    start = (0.0, v0, 0.0)     # every path's start state (reference :823-826) goes to the first stepping launch as constants

    def advance(i: int, forward: float, snap_row: int, qvar_row, spot_ptr: int) -> None:
        nb, dt = grids[i]
        eng.logsv_slice_rng(nb, dt, theta, kappa1, kappa2, beta, volvol, float(vol_backbone_etas[i]), is_spot_measure,
                            rng_seed, call_id, int(step0[i]), forward, snap_row, qvar_row, spot_ptr,
                            start=start if i == 0 else None)

    def advance_chain(need_qvar: bool, spot_ptr: int) -> None:
        eng.logsv_chain_rng([g[0] for g in grids], [g[1] for g in grids], vol_backbone_etas, forwards, theta, kappa1,
                            kappa2, beta, volvol, is_spot_measure, rng_seed, call_id, 0, need_qvar, spot_ptr, start=start)

    return price_chain_on_engine(eng, comm, nb_path, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms,
                                 variable_type, advance,
                                 advance_chain=advance_chain if (WHOLE_CHAIN_STEPPING and len(grids) > 1) else None)


def get_randoms_for_chain_valuation(ttms: np.ndarray, nb_path: int = 100000, nb_steps_per_year: int = 360,
                                    seed: int = 10) -> Tuple[List[np.ndarray], List[np.ndarray], List[float]]:
    """host-side fixed randoms, identical to the reference (:1051-1074): a local RandomState(seed) draws, per
    slice, W0 then W1 of shape [nb_steps_i, nb_path]; the global NumPy RNG is untouched."""
    rng = np.random.RandomState(seed)
    W0s, W1s, dts, t0 = [], [], [], 0.0
    for ttm in ttms:
        nb, dt = time_grid_steps(ttm=ttm - t0, nb_steps_per_year=nb_steps_per_year)
        W0s.append(rng.normal(0, 1, size=(nb, nb_path)))
        W1s.append(rng.normal(0, 1, size=(nb, nb_path)))
        dts.append(dt)
        t0 = ttm
    return W0s, W1s, dts


def upload_fixed_randoms(W0s: Sequence[np.ndarray], W1s: Sequence[np.ndarray], dts: Sequence[float], comm=None
                         ) -> DeviceRandoms:
    """copy the chain's fixed randoms to HBM once (this rank's path columns); pass the result as `W0s` to
    logsv_mc_chain_pricer_fixed_randoms.  This is what makes an MC calibration loop kernel-bound instead of
    PCIe-bound (reference logsv_pricer.py:520-527 draws the randoms once and re-prices per optimizer iterate)."""
    comm = comm or svdist.get_default_comm()
    offset, n_local = svdist.shard_range(np.asarray(W0s[0]).shape[1], comm.rank, comm.world)
    return DeviceRandoms(W0s, W1s, dts, n_local, offset)


def draw_fixed_randoms_on_device(ttms: np.ndarray, nb_path: int = 100000, nb_steps_per_year: int = 360, seed: int = 10,
                                 comm=None, in_hbm: bool = False) -> DeviceRandoms:
    """the device-side twin of get_randoms_for_chain_valuation + upload_fixed_randoms (no reference counterpart): the
    same per-expiry grids, the N(0,1) draws of the counter-based generator -- the draws logsv_mc_chain_pricer(seed=seed)
    consumes at its call 0 -- frozen.  By default NOTHING is stored: the object is the stream's definition (seed, call 0)
    and every pricing on it regenerates the same normals in registers (DeviceRandoms.frozen), bit-identical to
    logsv_mc_chain_pricer on that stream.  in_hbm=True materialises them in HBM instead (16 bytes per path-step, priced
    by the streamed-randoms kernels in the reference's evaluation order: the same draws, rounding-level different prices).
    Either way statistically equivalent to, not bit-identical with, the RandomState(seed) arrays of the reference."""
    comm = comm or svdist.get_default_comm()
    offset, n_local = svdist.shard_range(nb_path, comm.rank, comm.world)
    grids, t0 = [], 0.0
    for ttm in ttms:
        nb, dt = time_grid_steps(ttm=ttm - t0, nb_steps_per_year=nb_steps_per_year)
        grids.append((nb, dt))
        t0 = ttm
    get_engine(n_local, path_offset=offset)               # the library and the device are up before the first launch
    make = DeviceRandoms.drawn_on_device if in_hbm else DeviceRandoms.frozen
    return make([g[0] for g in grids], [g[1] for g in grids], nb_path, n_local, offset, seed)


def logsv_mc_chain_pricer_fixed_randoms(ttms: np.ndarray, forwards: np.ndarray, discfactors: np.ndarray,
                                        strikes_ttms: Sequence[np.ndarray], optiontypes_ttms: Sequence[np.ndarray],
                                        W0s: Sequence[np.ndarray], W1s: Sequence[np.ndarray], dts: Sequence[float],
                                        v0: float, theta: float, kappa1: float, kappa2: float, beta: float,
                                        volvol: float, vol_backbone_etas: np.ndarray, is_spot_measure: bool = True,
                                        variable_type: VariableType = VariableType.LOG_RETURN, comm=None,
                                        return_ivols: bool = False) -> Tuple[List[np.ndarray], ...]:
    """chain MC on supplied randoms (reference :1100-1162): nb_path = W0s[0].shape[1]; each rank uploads only
    its own column range of the host arrays.  W0s may instead be a DeviceRandoms (upload_fixed_randoms): the
    randoms then stay in HBM across calls and W1s / dts are taken from it."""
    variable_type_code(variable_type)
    comm = comm or svdist.get_default_comm()
    resident = W0s if isinstance(W0s, DeviceRandoms) else None
    nb_path = resident.nb_path if resident else np.asarray(W0s[0]).shape[1]
    offset, n_local = svdist.shard_range(nb_path, comm.rank, comm.world)
    if resident and (resident.n_local, resident.col0) != (n_local, offset):
        raise ValueError("DeviceRandoms were uploaded for a different path shard")
    if FUSED_FIXED_RANDOMS_DRIVER and resident and comm.world == 1 and len(resident) == len(ttms):
        # single GPU, randoms resident: the whole chain in one call of the fused C++ driver
        strikes = [np.ascontiguousarray(np.asarray(k, dtype=np.float64)) for k in strikes_ttms]
        codes = [option_type_codes(t) for t in optiontypes_ttms]
        out = resident.price_logsv_chain(ttms, forwards, discfactors, [k.ravel() for k in strikes],
                                         [c.ravel() for c in codes], v0, theta, kappa1, kappa2, beta, volvol,
                                         vol_backbone_etas, is_spot_measure, variable_type_code(variable_type),
                                         want_ivols=return_ivols)
        return tuple([_shaped_like(a, k) for a, k in zip(part, strikes_ttms)] for part in out)
    if resident and resident.is_frozen:
        # frozen randoms, sharded over ranks (or the fused driver switched off): the on-device-RNG chain on the stream the
        # object names -- the same numbers the fused route gives on one GPU
        fr_seed, fr_call = resident.frozen_stream
        prices, stderrs = _logsv_mc_chain_on_grids(list(zip(resident.nb_steps, resident.dts)), fr_seed, fr_call, comm, ttms,
                                                   forwards, discfactors, strikes_ttms, optiontypes_ttms, v0, theta, kappa1,
                                                   kappa2, beta, volvol, vol_backbone_etas, is_spot_measure, nb_path,
                                                   variable_type)
        if not return_ivols:
            return prices, stderrs
        return prices, stderrs, _host_ivols(prices, ttms, forwards, strikes_ttms, optiontypes_ttms, discfactors)
    eng = get_engine(n_local, path_offset=offset)
    eng.fill_state(0.0, v0, 0.0)

    def advance(i: int, forward: float, snap_row: int, qvar_row, spot_ptr: int) -> None:
        if resident:
            eng.logsv_slice_w(resident.nb_steps[i], resident.dts[i], theta, kappa1, kappa2, beta, volvol,
                              float(vol_backbone_etas[i]), is_spot_measure, resident.w0[i].ptr, resident.w1[i].ptr,
                              forward, snap_row, qvar_row, spot_ptr)
            return
        W0, W1 = np.asarray(W0s[i]), np.asarray(W1s[i])
        if W0.shape != W1.shape or W0.shape[1] != nb_path:
            raise ValueError("every W0/W1 must have shape [nb_steps_i, nb_path]")
        w0, w1 = eng.upload_randoms((W0, W1), col0=offset)
        eng.logsv_slice_w(W0.shape[0], float(dts[i]), theta, kappa1, kappa2, beta, volvol,
                          float(vol_backbone_etas[i]), is_spot_measure, w0, w1, forward, snap_row, qvar_row, spot_ptr)

    prices, stderrs = price_chain_on_engine(eng, comm, nb_path, ttms, forwards, discfactors, strikes_ttms,
                                            optiontypes_ttms, variable_type, advance)
    if not return_ivols:
        return prices, stderrs
    return prices, stderrs, _host_ivols(prices, ttms, forwards, strikes_ttms, optiontypes_ttms, discfactors)


def _shaped_like(a: np.ndarray, k) -> np.ndarray:
    """a result row in the shape of the strikes it belongs to (the reference returns arrays shaped like strikes_ttms[i])"""
    shape = k.shape if isinstance(k, np.ndarray) else np.shape(k)
    return a if a.shape == shape else a.reshape(shape)


def _host_ivols(prices, ttms, forwards, strikes_ttms, optiontypes_ttms, discfactors) -> List[np.ndarray]:
    from ..data.option_chain import black_ivols_native           # the host twin of the graph's implied-vol kernel
    return [black_ivols_native(np.asarray(p, dtype=float).ravel(), float(t), float(f), np.asarray(k, dtype=float).ravel(),
                               np.asarray(ty).ravel(), float(d)).reshape(np.shape(k))
            for p, t, f, k, ty, d in zip(prices, ttms, forwards, strikes_ttms, optiontypes_ttms, discfactors)]


def logsv_mc_chain_pricer_fixed_randoms_batch(params_list: Sequence[LogSvParams], ttms: np.ndarray, forwards: np.ndarray,
                                              discfactors: np.ndarray, strikes_ttms: Sequence[np.ndarray],
                                              optiontypes_ttms: Sequence[np.ndarray], W0s: DeviceRandoms,
                                              is_spot_measure: bool = True,
                                              variable_type: VariableType = VariableType.LOG_RETURN,
                                              return_ivols: bool = False, comm=None) -> List[Tuple[List[np.ndarray], ...]]:
    """logsv_mc_chain_pricer_fixed_randoms for SEVERAL parameter sets on the same resident randoms (W0s: the result of
    upload_fixed_randoms / draw_fixed_randoms_on_device): per set the (prices, stderrs[, ivols]) of a single call, the
    same bits.  On one GPU, 2..8 sets go through ONE replayed graph whose stepping launch reads the randoms once for all
    sets (svmc_logsv_chain_price_fixed_sets) -- the base point of an SLSQP iterate and its finite-difference neighbours;
    otherwise the sets are priced one after the other.  Not in the reference API."""
    vt = variable_type_code(variable_type)
    comm = comm or svdist.get_default_comm()
    if not isinstance(W0s, DeviceRandoms):
        raise TypeError("the batched pricer works on resident randoms (upload_fixed_randoms / draw_fixed_randoms_on_device)")
    codes = [option_type_codes(t) for t in optiontypes_ttms]
    if comm.world == 1 and FUSED_FIXED_RANDOMS_DRIVER and len(W0s) == len(ttms):
        strikes = [np.ascontiguousarray(np.asarray(k, dtype=np.float64)) for k in strikes_ttms]
        rows = np.ones((len(params_list), 6 + len(ttms)))
        for row, p in zip(rows, params_list):
            row[:6] = (p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta, p.volvol)
            if p.vol_backbone is not None:
                row[6:] = p.get_vol_backbone_etas(ttms=ttms)
        out = []
        for q0 in range(0, len(params_list), 8):                      # a launch takes up to 8 sets
            out += W0s.price_logsv_chain_sets(ttms, forwards, discfactors, [k.ravel() for k in strikes],
                                              [c.ravel() for c in codes], rows[q0:q0 + 8], is_spot_measure, vt,
                                              want_ivols=return_ivols)
        if all(np.ndim(k) == 1 for k in strikes_ttms):
            return out
        return [tuple([_shaped_like(a, k) for a, k in zip(part, strikes_ttms)] for part in res) for res in out]
    return [logsv_mc_chain_pricer_fixed_randoms(ttms=ttms, forwards=forwards, discfactors=discfactors,
                                                strikes_ttms=strikes_ttms, optiontypes_ttms=optiontypes_ttms, W0s=W0s,
                                                W1s=None, dts=None, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1,
                                                kappa2=p.kappa2, beta=p.beta, volvol=p.volvol,
                                                vol_backbone_etas=p.get_vol_backbone_etas(ttms=ttms),
                                                is_spot_measure=is_spot_measure, variable_type=variable_type, comm=comm,
                                                return_ivols=return_ivols) for p in params_list]


# ---------------------------------------------------------------------------------------------------
# rough LogSV (Markovian lift of the fractional kernel), reference :1075-1232
# ---------------------------------------------------------------------------------------------------
def get_randoms_for_rough_vol_chain_valuation(ttms: np.ndarray, nb_path: int = 100000, nb_steps_per_year: int = 360,
                                              seed: int = 10) -> Tuple[np.ndarray, np.ndarray, List[np.ndarray]]:
    """host randoms and per-expiry time grids, identical to the reference (:1075-1097): every expiry has its own
    grid from 0 to T_i with int(T_i*spy)+1 steps; Z0 then Z1 of shape [nb_steps_last, nb_path] from a local
    RandomState(seed)."""
    rng = np.random.RandomState(seed)
    grids = [set_time_grid(ttm=ttm, nb_steps_per_year=nb_steps_per_year)[2] for ttm in ttms]
    nb_last = grids[-1].size - 1
    Z0 = rng.normal(0, 1, size=(nb_last, nb_path))
    Z1 = rng.normal(0, 1, size=(nb_last, nb_path))
    return Z0, Z1, grids


def upload_rough_randoms(Z0: np.ndarray, Z1: np.ndarray, comm=None) -> DeviceRandoms:
    """copy the rough chain's Z0 / Z1 to HBM once (this rank's path columns); pass the result as `Z0` to
    rough_logsv_mc_chain_pricer_fixed_randoms"""
    comm = comm or svdist.get_default_comm()
    offset, n_local = svdist.shard_range(np.asarray(Z0).shape[1], comm.rank, comm.world)
    return DeviceRandoms([Z0], [Z1], [0.0], n_local, offset)


def _rough_finalize(normalize_stderr: bool):
    """The reference passes a [1, nb_path] log-spot to compute_mc_vars_payoff, whose "/ sqrt(x0.shape[0])"
    (utils/mc_payoffs.py:88) then divides by 1: its second return is the payoff's standard deviation.  Reproduced
    by default; normalize_stderr=True returns the standard error instead."""
    if normalize_stderr:
        return payoff_finalize

    def finalize(sums, shifts, discfactor, n_path_total):
        p, e = payoff_finalize(sums, shifts, discfactor, n_path_total)
        return p, e * np.sqrt(n_path_total)
    return finalize


def _rough_coefficients(sigma0, beta, orthog_vol, weights, nodes):
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    if not (weights.ndim == 1 and weights.shape == nodes.shape):
        raise AssertionError("weights and nodes must be 1-d arrays of one shape")      # reference :1188
    if not 1 <= nodes.size <= 3:
        raise NotImplementedError("the Markovian lift is built for 1 to 3 factors (LogSvParams.approximate_kernel)")
    v0 = np.full(nodes.size, sigma0 / np.sum(weights))
    volvol = float(np.sqrt(beta ** 2 + orthog_vol ** 2))
    return weights, nodes, v0, float(beta / volvol), volvol


def rough_logsv_mc_chain_pricer_fixed_randoms(ttms: np.ndarray, forwards: np.ndarray, discfactors: np.ndarray,
                                              strikes_ttms: Sequence[np.ndarray],
                                              optiontypes_ttms: Sequence[np.ndarray], Z0: np.ndarray, Z1: np.ndarray,
                                              sigma0: float, theta: float, kappa1: float, kappa2: float, beta: float,
                                              orthog_vol: float, weights: np.ndarray, nodes: np.ndarray,
                                              timegrids: Sequence[np.ndarray],
                                              variable_type: VariableType = VariableType.LOG_RETURN,
                                              debug: bool = False, comm=None, normalize_stderr: bool = False
                                              ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """rough-LogSV chain on supplied N(0,1) draws (reference :1164-1232).  Z0 may be the result of
    upload_rough_randoms (Z1 is then ignored); otherwise Z0/Z1 go to HBM once per call (this rank's path columns); every expiry is then one kernel that re-simulates from time 0 over the first len(timegrid_i)-1 rows
    with step timegrid_i[1]-timegrid_i[0], exactly as the reference does."""
    variable_type_code(variable_type)
    weights, nodes, v0, rho, volvol = _rough_coefficients(sigma0, beta, orthog_vol, weights, nodes)
    comm = comm or svdist.get_default_comm()
    resident = Z0 if isinstance(Z0, DeviceRandoms) else None
    if resident is None:
        Z0, Z1 = np.asarray(Z0), np.asarray(Z1)
        if Z0.ndim != 2 or Z0.shape != Z1.shape:
            raise ValueError("Z0 and Z1 must both have shape [nb_steps, nb_path]")
    nb_path = resident.nb_path if resident else Z0.shape[1]
    nb_rows = resident.nb_steps[0] if resident else Z0.shape[0]
    nbs = [int(np.asarray(g).size) - 1 for g in timegrids]
    if len(nbs) != len(ttms) or max(nbs) > nb_rows or min(nbs) < 1:
        raise ValueError("one time grid per maturity, each with at most Z0.shape[0] steps")
    offset, n_local = svdist.shard_range(nb_path, comm.rank, comm.world)
    if resident and (resident.n_local, resident.col0) != (n_local, offset):
        raise ValueError("DeviceRandoms were uploaded for a different path shard")
    eng = get_engine(n_local, path_offset=offset)
    if resident:
        z0, z1 = resident.w0[0].ptr, resident.w1[0].ptr
    else:
        z0, z1 = eng.upload_randoms((Z0[:max(nbs)], Z1[:max(nbs)]), col0=offset)

    def advance(i: int, forward: float, snap_row: int, qvar_row, spot_ptr: int) -> None:
        h = float(timegrids[i][1] - timegrids[i][0])
        eng.rough_logsv(nbs[i], h, nodes, weights, v0, theta, kappa1, kappa2, rho, volvol, z0, z1,
                        slice_out=(forward, snap_row, qvar_row, spot_ptr))
        if debug:
            x, _, _ = eng.get_state()
            vw = weights @ eng.get_factors(nodes.size)
            print(f"Number of paths with negative vol: {np.sum(vw < 0.0)}, nan vol: {np.count_nonzero(np.isnan(vw))}")
            print(f"Mean spot Strand: {np.mean(np.exp(x))}, nan spots: {np.count_nonzero(np.isnan(x))}")

    def advance_chain(need_q: bool, spot_ptr: int) -> None:
        # the expiries are independent simulations from time 0 (:1206-1216): all of them in one launch, side by side
        eng.rough_logsv_chain(nbs, [float(g[1] - g[0]) for g in timegrids], forwards, nodes, weights, v0, theta, kappa1, kappa2,
                              rho, volvol, need_q, spot_ptr, z0_ptr=z0, z1_ptr=z1)

    one_launch = ROUGH_CHAIN_ONE_LAUNCH and not debug and len(nbs) <= 16 and hasattr(eng, "rough_logsv_chain")
    return price_chain_on_engine(eng, comm, nb_path, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms,
                                 variable_type, advance, finalize=_rough_finalize(normalize_stderr),
                                 advance_chain=advance_chain if one_launch else None)


def rough_logsv_mc_chain_pricer(ttms: np.ndarray, forwards: np.ndarray, discfactors: np.ndarray,
                                strikes_ttms: Sequence[np.ndarray], optiontypes_ttms: Sequence[np.ndarray],
                                sigma0: float, theta: float, kappa1: float, kappa2: float, beta: float,
                                orthog_vol: float, weights: np.ndarray, nodes: np.ndarray, nb_path: int = 100000,
                                nb_steps_per_year: int = 360, variable_type: VariableType = VariableType.LOG_RETURN,
                                seed: Optional[int] = None, comm=None, normalize_stderr: bool = False
                                ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """addition (no reference counterpart): the same chain with the N(0,1) draws generated in the kernel (stream 3 of
    the counter-based generator), so nothing but the chain itself crosses PCIe.  Expiry i uses the draws of steps
    0..nb_steps_i-1, i.e. the same nesting of randoms over expiries as the fixed-randoms pricer."""
    variable_type_code(variable_type)
    weights, nodes, v0, rho, volvol = _rough_coefficients(sigma0, beta, orthog_vol, weights, nodes)
    comm = comm or svdist.get_default_comm()
    offset, n_local = svdist.shard_range(nb_path, comm.rank, comm.world)
    eng = get_engine(n_local, path_offset=offset)
    rng_seed, call_id = next_rng_call(seed)
    grids = [set_time_grid(ttm=ttm, nb_steps_per_year=nb_steps_per_year)[:2] for ttm in ttms]

    def advance(i: int, forward: float, snap_row: int, qvar_row, spot_ptr: int) -> None:
        nb, h = grids[i]
        eng.rough_logsv(nb, h, nodes, weights, v0, theta, kappa1, kappa2, rho, volvol, seed=rng_seed, call_id=call_id,
                        slice_out=(forward, snap_row, qvar_row, spot_ptr))

    def advance_chain(need_q: bool, spot_ptr: int) -> None:
        eng.rough_logsv_chain([g[0] for g in grids], [g[1] for g in grids], forwards, nodes, weights, v0, theta, kappa1, kappa2,
                              rho, volvol, need_q, spot_ptr, seed=rng_seed, call_id=call_id)

    one_launch = ROUGH_CHAIN_ONE_LAUNCH and len(grids) <= 16 and hasattr(eng, "rough_logsv_chain")
    return price_chain_on_engine(eng, comm, nb_path, ttms, forwards, discfactors, strikes_ttms, optiontypes_ttms,
                                 variable_type, advance, finalize=_rough_finalize(normalize_stderr),
                                 advance_chain=advance_chain if one_launch else None)
