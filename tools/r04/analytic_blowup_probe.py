"""How long the analytic chain pricer takes on parameter vectors whose coefficient ODEs blow up before the expiry (a
calibrator's line search can wander there), and what it hands back.  Run under `timeout`:
    timeout 120 python tools/r04/analytic_blowup_probe.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import stochvolmodels_amd as sv

kk = np.linspace(0.6, 1.4, 21)
ty = np.where(kk >= 1.0, "C", "P")
pricer = sv.LogSVPricer()
cases = {"btc": sv.LOGSV_BTC_PARAMS}
for vv, ttm in ((5.0, 2.0), (20.0, 2.0), (50.0, 5.0), (200.0, 10.0)):
    cases[f"volvol={vv} beta={vv} ttm={ttm}"] = (sv.LogSvParams(sigma0=1.0, theta=1.0, kappa1=0.1, kappa2=0.1, beta=vv, volvol=vv), ttm)
cases["sigma0=50"] = (sv.LogSvParams(sigma0=50.0, theta=50.0, kappa1=0.01, kappa2=0.0, beta=-10.0, volvol=30.0), 10.0)
for name, c in cases.items():
    p, ttm = c if isinstance(c, tuple) else (c, 1.0)
    chain = sv.OptionChain(ttms=np.array([0.25 * ttm, ttm]), forwards=np.ones(2), strikes_ttms=(kk,) * 2, optiontypes_ttms=(ty,) * 2, ids=None)
    for vt in ("LOG_RETURN",):
        t0 = time.perf_counter()
        try:
            out = pricer.price_chain(chain, p, variable_type=getattr(sv.VariableType, vt))
            arr = np.stack(out)
            res = {"nan": int(np.isnan(arr).sum()), "finite": int(np.isfinite(arr).sum()), "head": [float(v) for v in arr[-1][:3]]}
        except Exception as e:                                  # noqa: BLE001
            res = {"error": repr(e)[:200]}
        res.update(case=name, variable=vt, seconds=round(time.perf_counter() - t0, 4))
        print(json.dumps(res), flush=True)
