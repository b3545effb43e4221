"""One MC calibration objective evaluation (4 x 13 chain, 10^5 paths x 364 steps on resident randoms, implied vols from the
graph) repeated: run under `rocprofv3 --kernel-trace --hip-trace --stats` to see where the 0.2 ms go (device kernels vs the
graph launch vs the synchronisation).  Prints the host-side median per call."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import stochvolmodels_amd as sv
from stochvolmodels_amd.engine import option_type_codes
from stochvolmodels_amd.pricers import logsv_pricer as lp

nb_path = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
k = np.linspace(0.7, 1.3, 13)
ty = np.where(k >= 1.0, "C", "P")
p = sv.LOGSV_BTC_PARAMS
res = lp.draw_fixed_randoms_on_device(ttms, nb_path=nb_path, nb_steps_per_year=360, seed=10) if hasattr(lp, "draw_fixed_randoms_on_device") \
    else lp.upload_fixed_randoms(*lp.get_randoms_for_chain_valuation(ttms, nb_path=nb_path, nb_steps_per_year=360, seed=10))
codes = [option_type_codes(ty)] * 4


def call(iv):
    return res.price_logsv_chain(ttms, np.ones(4), np.ones(4), [k] * 4, codes, p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta,
                                 p.volvol, np.ones(4), True, 1, use_graph=True, want_ivols=iv)


out = {}
for iv in (False, True):
    for _ in range(20):
        call(iv)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call(iv)
        ts.append(time.perf_counter() - t0)
    out["with_ivols" if iv else "prices_only"] = {"median_ms": 1e3 * float(np.median(ts)), "min_ms": 1e3 * float(np.min(ts))}
print(json.dumps(out))
