"""
Payoff codes and state-variable selectors of the Monte Carlo path
(mirror of the reference's utils/config.py:8-23; values are part of the drop-in contract).
"""
from enum import Enum


class OptionType(str, Enum):
    CALL = "C"
    PUT = "P"
    INVERSE_CALL = "IC"
    INVERSE_PUT = "IP"


class VariableType(Enum):
    LOG_RETURN = 1
    Q_VAR = 2
    SIGMA = 3
