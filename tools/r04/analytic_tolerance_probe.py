"""The analytic LogSV chain (SURVEY a11, C5's analytic side) at looser tolerances of the coefficient ODE: time per 4 x 21 chain and
the distance to (a) the committed tolerance's prices, (b) the reference as shipped (its RK45 at rtol 1e-3: tests/golden/analytic.npz).
The reference itself solves at rtol 1e-3; the committed 1e-10 isolates the algebra from the solver."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import stochvolmodels_amd as sv
from stochvolmodels_amd import analytic

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "analytic.npz"))
kk, types, ttms = g["strikes"], g["types"], g["ttms"]
one = np.ones(4)
chain = sv.OptionChain(ttms=ttms, forwards=one, strikes_ttms=(kk,) * 4, optiontypes_ttms=(types,) * 4, ids=None)
pricer = sv.LogSVPricer()
sets = {}
for tag in ("btc", "readme", "quick", "test", "fig3"):
    v = [float(a) for a in g[f"logsv_{tag}_params"]]
    sets[tag] = sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5])
base = {}
for rtol, atol in ((1e-10, 1e-12), (1e-8, 1e-10), (1e-6, 1e-8), (1e-5, 1e-7), (1e-4, 1e-6), (1e-3, 1e-6)):
    analytic.ODE_RTOL, analytic.ODE_ATOL = rtol, atol
    worst_self, worst_ref, ms = 0.0, 0.0, []
    for tag, p in sets.items():
        pr = np.stack(pricer.price_chain(chain, p))
        for _ in range(3):
            pricer.price_chain(chain, p)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            pricer.price_chain(chain, p)
            ts.append(time.perf_counter() - t0)
        ms.append(1e3 * float(np.median(ts)))
        base.setdefault(tag, pr)
        worst_self = max(worst_self, float(np.max(np.abs(pr - base[tag]))))
        worst_ref = max(worst_ref, float(np.max(np.abs(pr - g[f"logsv_{tag}_prices"]))))
    print(json.dumps(dict(rtol=rtol, atol=atol, ms_per_chain_by_set=[round(m, 3) for m in ms],
                          max_abs_dev_from_rtol_1e10=worst_self, max_abs_dev_from_reference_as_shipped=worst_ref)))
