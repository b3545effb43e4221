#!/usr/bin/env python
"""
Where one calibration objective evaluation spends its time: BTC-style 4-expiry chain, MC engine on
resident fixed randoms.  Prints one JSON object per line.

    python tools/bench_calibration.py [nb_path]
"""
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd.pricers import logsv_pricer as lp  # noqa: E402


def main():
    nb_path = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    chain0 = sv.OptionChain(ttms=ttms, forwards=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4,
                            discfactors=np.ones(4), ids=np.array(list("abcd")))
    p = sv.LOGSV_BTC_PARAMS
    from stochvolmodels_amd.engine import get_engine
    get_engine(1024).synchronize()            # library load + HIP runtime start-up are not part of the upload time
    t0 = time.perf_counter()
    W = lp.get_randoms_for_chain_valuation(ttms, nb_path=nb_path, nb_steps_per_year=360, seed=10)
    t_draw = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = lp.upload_fixed_randoms(*W)
    t_up = time.perf_counter() - t0
    kw = dict(ttms=ttms, forwards=chain0.forwards, discfactors=chain0.discfactors, strikes_ttms=chain0.strikes_ttms,
              optiontypes_ttms=chain0.optiontypes_ttms, v0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2,
              beta=p.beta, volvol=p.volvol, vol_backbone_etas=np.ones(4))

    def price():
        return lp.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, **kw)[0]

    def ivols(prices):
        return chain0.compute_model_ivols_from_chain_data(model_prices=prices)

    def timed(fn, n=200):
        """median wall time of one call: the ROCm runtime stalls the process once for tens of ms a fixed time after
        start-up (profiles/r02_runtime_stall.txt), which a mean over a short loop smears over whatever runs then"""
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def price_ivols():
        return lp.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, return_ivols=True, **kw)[2]

    sets = [sv.LogSvParams(sigma0=p.sigma0 + 1e-8 * j, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta,
                           volvol=p.volvol) for j in range(6)]
    chain_kw = {k_: kw[k_] for k_ in ("ttms", "forwards", "discfactors", "strikes_ttms", "optiontypes_ttms")}

    def price_six_sets():
        return lp.logsv_mc_chain_pricer_fixed_randoms_batch(params_list=sets, W0s=res, return_ivols=True, **chain_kw)

    pr = price()
    t_price = timed(price)
    t_six = timed(price_six_sets)
    t_price_iv = timed(price_ivols)
    t_iv = timed(lambda: ivols(pr))
    steps = sum(res.nb_steps)
    lp.FUSED_FIXED_RANDOMS_DRIVER = False
    t_price_py = timed(price)
    lp.FUSED_FIXED_RANDOMS_DRIVER = True
    from stochvolmodels_amd.engine import option_type_codes
    direct = lambda g: res.price_logsv_chain(ttms, chain0.forwards, chain0.discfactors, [k] * 4,       # noqa: E731
                                             [option_type_codes(ty)] * 4, p.sigma0, p.theta, p.kappa1, p.kappa2, p.beta,
                                             p.volvol, np.ones(4), True, 1, use_graph=g)
    t_direct = {}
    for g in (False, True):
        t_direct[g] = timed(lambda: direct(g))
    print(json.dumps(dict(nb_path=nb_path, steps=steps, host_draw_s=t_draw, upload_s=t_up, price_ms=1e3 * t_price, price_with_ivols_from_the_graph_ms=1e3 * t_price_iv, six_parameter_sets_one_replay_with_ivols_ms=1e3 * t_six, price_python_driver_ms=1e3 * t_price_py, fused_no_graph_ms=1e3 * t_direct[False],
                          fused_graph_ms=1e3 * t_direct[True],
                          ivol_ms=1e3 * t_iv, kernel_floor_ms=1e3 * nb_path * steps / 3.7e11)))
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(20):
        ivols(price())
    prof.disable()
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(18)
    print(s.getvalue()[:3500])


if __name__ == "__main__":
    main()
