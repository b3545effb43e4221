#!/usr/bin/env python
"""
A whole SLSQP calibration with the MC engine: 4 x 13 chain whose "market" vols are the analytic model's at
a known parameter set, PARAMS5 from a displaced start, 10^5 paths x 364 steps of fixed randoms -- once with the
reference's randoms (RandomState arrays drawn on the host, uploaded once) and once with the randoms drawn in HBM
(device_randoms=True).  Every objective evaluation is one replay of the six-node graph that ends with the implied vols.
One JSON line.

    python tools/bench_calibration_mc.py [nb_path]
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
    nb_path = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    base = dict(ttms=ttms, forwards=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4, discfactors=np.ones(4),
                ids=np.array(list("abcd")))
    truth = sv.LogSvParams(sigma0=0.85, theta=1.0, kappa1=3.0, kappa2=3.0, beta=0.15, volvol=1.6)
    vols = sv.LogSVPricer().compute_model_ivols_for_chain(option_chain=sv.OptionChain(**base), params=truth)
    chain = sv.OptionChain(bid_ivs=tuple(v - 0.002 for v in vols), ask_ivs=tuple(v + 0.002 for v in vols), **base)
    start = sv.LogSvParams(sigma0=0.7, theta=0.8, kappa1=3.0, kappa2=3.0, beta=0.0, volvol=1.2)
    out = dict(nb_path=nb_path)
    for name, dev, batched in (("host_randoms", False, True), ("device_randoms", True, True), ("host_randoms_2", False, True),
                               ("device_randoms_2", True, True), ("device_randoms_slsqp_differences", True, False),
                               ("device_randoms_slsqp_differences_2", True, False)):
        p = sv.LogSVPricer()
        t0 = time.perf_counter()
        fit = p.calibrate_model_params_to_chain(option_chain=chain, params0=start, disp=False, nb_path=nb_path, nb_steps=360,
                                                seed=10, calibration_engine=sv.CalibrationEngine.MC,
                                                model_calibration_type=sv.LogsvModelCalibrationType.PARAMS5,
                                                device_randoms=dev, batched_gradient=batched)
        dt = time.perf_counter() - t0
        out[name] = dict(seconds=dt, n_eval=p.last_calibration["n_eval"], objective=p.last_calibration["objective"],
                         fit=[fit.sigma0, fit.theta, fit.kappa1, fit.kappa2, fit.beta, fit.volvol])
    out["speedup"] = out["host_randoms_2"]["seconds"] / out["device_randoms_2"]["seconds"]
    out["speedup_of_the_batched_gradient"] = (out["device_randoms_slsqp_differences_2"]["seconds"] /
                                              out["device_randoms_2"]["seconds"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
