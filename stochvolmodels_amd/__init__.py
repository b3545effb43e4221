"""
stochvolmodels_amd -- MI355X-native Monte Carlo engine for the StochVolModels hot path.

Module layout and public names mirror the reference package (`stochvolmodels`) for the Monte Carlo path:
    stochvolmodels_amd.pricers.logsv_pricer   LogSVPricer, logsv_mc_chain_pricer(_fixed_randoms),
                                              simulate_logsv_x_vol_terminal, get_randoms_for_chain_valuation
    stochvolmodels_amd.pricers.heston_pricer  HestonPricer, HestonParams, heston_mc_chain_pricer,
                                              simulate_heston_x_vol_terminal
    stochvolmodels_amd.utils.mc_payoffs       compute_mc_vars_payoff
    stochvolmodels_amd.utils.funcs            set_time_grid, set_seed, timer
    stochvolmodels_amd.utils.config           OptionType, VariableType
    stochvolmodels_amd.data.option_chain      OptionChain
All arithmetic runs in libsvmc.so (hand-written HIP for gfx950, C ABI in include/svmc.h); there is no CPU
fallback.  Names resolve lazily so importing the package does not load the GPU library.
"""
import importlib

__version__ = "0.3.0"
# the counter-based random stream the generators draw from (include/svmc.h SVMC_RNG_STREAM_VERSION, CHANGELOG.md): results
# for a given seed are reproducible within one stream version
RNG_STREAM_VERSION = 4

_EXPORTS = {
    "OptionType": "utils.config", "VariableType": "utils.config",
    "set_time_grid": "utils.funcs", "set_seed": "utils.funcs", "timer": "utils.funcs",
    "to_flat_np_array": "utils.funcs",
    "compute_mc_vars_payoff": "utils.mc_payoffs",
    "OptionChain": "data.option_chain",
    "ModelParams": "pricers.model_pricer", "ModelPricer": "pricers.model_pricer",
    "LogSvParams": "pricers.logsv.logsv_params",
    "LogSVPricer": "pricers.logsv_pricer", "LOGSV_BTC_PARAMS": "pricers.logsv_pricer",
    "logsv_mc_chain_pricer": "pricers.logsv_pricer",
    "logsv_mc_chain_pricer_fixed_randoms": "pricers.logsv_pricer",
    "simulate_logsv_x_vol_terminal": "pricers.logsv_pricer",
    "simulate_vol_paths": "pricers.logsv_pricer",
    "vol_path_moments": "pricers.logsv_pricer",
    "get_randoms_for_chain_valuation": "pricers.logsv_pricer",
    "upload_fixed_randoms": "pricers.logsv_pricer",
    "draw_fixed_randoms_on_device": "pricers.logsv_pricer",
    "logsv_mc_chain_pricer_fixed_randoms_batch": "pricers.logsv_pricer",
    "get_randoms_for_rough_vol_chain_valuation": "pricers.logsv_pricer",
    "rough_logsv_mc_chain_pricer_fixed_randoms": "pricers.logsv_pricer",
    "rough_logsv_mc_chain_pricer": "pricers.logsv_pricer",
    "upload_rough_randoms": "pricers.logsv_pricer",
    "LogsvModelCalibrationType": "pricers.logsv_pricer",
    "ConstraintsType": "pricers.logsv_pricer",
    "CalibrationEngine": "pricers.logsv_pricer",
    "CalibrationError": "utils.calibration",
    "logsv_chain_pricer": "pricers.logsv_pricer", "set_vol_scaler": "pricers.logsv_pricer",
    "logsv_chain_pricer_batch": "pricers.logsv_pricer",
    "ExpansionOrder": "pricers.logsv.affine_expansion", "compute_logsv_a_mgf_grid": "pricers.logsv.affine_expansion",
    "heston_chain_pricer": "pricers.heston_pricer", "compute_heston_mgf_grid": "pricers.heston_pricer",
    "HestonPricer": "pricers.heston_pricer", "HestonParams": "pricers.heston_pricer",
    "BTC_HESTON_PARAMS": "pricers.heston_pricer", "heston_mc_chain_pricer": "pricers.heston_pricer",
    "simulate_heston_x_vol_terminal": "pricers.heston_pricer",
    "compute_analytic_qvar": "pricers.logsv.vol_moments_ode",
    "compute_analytic_vol_moments": "pricers.logsv.vol_moments_ode",
    "fit_model_vol_backbone_to_varswaps": "pricers.logsv.vol_moments_ode",
    "compute_var_swap_strike": "utils.var_swap_pricer",
}

__all__ = sorted(_EXPORTS)


def __getattr__(name):
    mod = _EXPORTS.get(name)
    if mod is None:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
    value = getattr(importlib.import_module(f"{__name__}.{mod}"), name)
    globals()[name] = value
    return value


def __dir__():
    return sorted(list(globals()) + list(_EXPORTS))
