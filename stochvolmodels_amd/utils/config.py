"""
Payoff codes and state-variable selectors of the Monte Carlo path.  Names and values are the drop-in contract with
the reference (utils/config.py:8-23); the int8 codes the kernels take (include/svmc.h SVMC_CALL .. SVMC_INV_PUT,
SVMC_LOG_RETURN / SVMC_Q_VAR) are derived from them in engine.option_type_codes / mc_chain.variable_type_code.
"""
import enum

# payoff code strings as they appear in OptionChain.optiontypes_ttms; a str-enum so that "C" == OptionType.CALL
OptionType = enum.Enum("OptionType", [("CALL", "C"), ("PUT", "P"), ("INVERSE_CALL", "IC"), ("INVERSE_PUT", "IP")],
                       type=str, module=__name__)

# which simulated variable the payoff is written on: 1 -> spot from the log-return, 2 -> annualised quadratic variance,
# 3 -> volatility (raises NotImplementedError on the Monte Carlo path, as in the reference)
VariableType = enum.Enum("VariableType", [("LOG_RETURN", 1), ("Q_VAR", 2), ("SIGMA", 3)], module=__name__)
