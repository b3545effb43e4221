#!/usr/bin/env python
"""One frozen-randoms objective evaluation (one set, 4 x 13 chain) against the path count and the steps per year: where the
time is fixed cost and where it is stepping.  One JSON line: ms[paths][steps]."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4)
    q = sv.LOGSV_BTC_PARAMS
    out = {}
    for n in (16384, 65536, 100000, 131072, 262144, 524288):
        row = {}
        for spy in (90, 360, 1440):
            res = sv.draw_fixed_randoms_on_device(ttms, nb_path=n, nb_steps_per_year=spy, seed=10)
            fn = lambda: sv.logsv_mc_chain_pricer_fixed_randoms(  # noqa: E731
                W0s=res, W1s=None, dts=None, v0=q.sigma0, theta=q.theta, kappa1=q.kappa1, kappa2=q.kappa2, beta=q.beta,
                volvol=q.volvol, vol_backbone_etas=np.ones(4), return_ivols=True, **chain)
            fn(); fn()
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            row[str(int(sum(res.nb_steps)))] = round(1e3 * float(np.median(ts)), 4)
            res.free()
        out[str(n)] = row
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
