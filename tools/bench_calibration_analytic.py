#!/usr/bin/env python
"""
A whole SLSQP calibration with the ANALYTIC engine (the reference's default engine; SURVEY row f.3 on top of f.1): 4 x 13
chain whose "market" vols are the model's at a known parameter set, PARAMS5 from a displaced start -- once with SLSQP
differencing the objective itself (n + 1 chain pricings per iterate, one after the other) and once with the gradient's
bumped vectors priced in ONE batch of launches per iterate (LogSVPricer.price_chain_batch).  One JSON line.

    python tools/bench_calibration_analytic.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402


def main():
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    base = dict(ttms=ttms, forwards=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4, discfactors=np.ones(4),
                ids=np.array(list("abcd")))
    truth = sv.LogSvParams(sigma0=0.85, theta=1.0, kappa1=3.0, kappa2=3.0, beta=0.15, volvol=1.6)
    pricer = sv.LogSVPricer()
    chain0 = sv.OptionChain(**base)
    vols = pricer.compute_model_ivols_for_chain(option_chain=chain0, params=truth)
    chain = sv.OptionChain(bid_ivs=tuple(v - 0.002 for v in vols), ask_ivs=tuple(v + 0.002 for v in vols), **base)
    start = sv.LogSvParams(sigma0=0.7, theta=0.8, kappa1=3.0, kappa2=3.0, beta=0.0, volvol=1.2)
    out = {}
    for name, batched in (("slsqp_differences", False), ("batched_gradient", True), ("slsqp_differences_2", False),
                          ("batched_gradient_2", True)):
        p = sv.LogSVPricer()
        t0 = time.perf_counter()
        fit = p.calibrate_model_params_to_chain(option_chain=chain, params0=start, disp=False,
                                                calibration_engine=sv.CalibrationEngine.ANALYTIC,
                                                model_calibration_type=sv.LogsvModelCalibrationType.PARAMS5,
                                                batched_gradient=batched)
        dt = time.perf_counter() - t0
        out[name] = dict(seconds=dt, n_eval=p.last_calibration["n_eval"], gradient_batches=p.last_calibration["n_gradient_batches"],
                         objective=p.last_calibration["objective"],
                         fit=[fit.sigma0, fit.theta, fit.kappa1, fit.kappa2, fit.beta, fit.volvol])
    out["speedup"] = out["slsqp_differences_2"]["seconds"] / out["batched_gradient_2"]["seconds"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
