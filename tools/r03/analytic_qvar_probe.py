"""time of the analytic quadratic-variance chain (2 expiries, 40 000-point psi grid) -- the long-grid case of the coefficient-ODE
kernel; SVMC_MGF_ROW_MAX_POINTS=0 forces one lane per grid point, a large value forces 16-lane rows"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import stochvolmodels_amd as sv
kk = np.linspace(0.05, 0.5, 8)
chain = sv.OptionChain(ttms=np.array([0.25, 0.5]), forwards=np.ones(2), strikes_ttms=(kk, kk),
                       optiontypes_ttms=(np.array(["C"] * 8),) * 2, ids=None, discfactors=np.ones(2))
pricer, P = sv.LogSVPricer(), sv.LOGSV_BTC_PARAMS
for _ in range(3):
    pr = pricer.price_chain(chain, P, variable_type=sv.VariableType.Q_VAR)
t = time.perf_counter()
for _ in range(10):
    pr = pricer.price_chain(chain, P, variable_type=sv.VariableType.Q_VAR)
print({"row_max_points": os.environ.get("SVMC_MGF_ROW_MAX_POINTS", "default (8192)"), "ms_per_chain": round((time.perf_counter() - t) / 10 * 1e3, 3),
       "atm": float(pr[0][3])})
