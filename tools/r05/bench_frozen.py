#!/usr/bin/env python
"""
One MC calibration objective evaluation WITHOUT resident randoms: the 4 x 13 chain of
tools/bench_calibration.py, 10^5 paths x 364 steps.  Times, per call (median of n, host clock around the call):
  frozen      draw_fixed_randoms_on_device()            -> svmc_logsv_chain_price_frozen_sets (randoms regenerated in registers)
  hbm         draw_fixed_randoms_on_device(in_hbm=True) -> svmc_logsv_chain_price_fixed_iv / _fixed_sets (round 4: 582 MB resident)
for 1 set (an objective evaluation, prices + implied vols) and 6, 7 and 8 sets (the base point and the bumped vectors of a
finite-difference gradient), and checks that the frozen route's prices are logsv_mc_chain_pricer(seed)'s bit for bit.

    python tools/r05/bench_frozen.py [nb_path] [calls]            one JSON line
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402


def clock_mhz_of_last_launch():
    """the shader clock inside this thread's latest armed stepping launch (svmc_clock_probe_arm / _read)"""
    import ctypes as C
    from stochvolmodels_amd import _lib
    st = (C.c_uint64 * 8)()
    _lib.check(_lib.load().svmc_clock_probe_read(st, None))
    t0, r0, t1, r1 = st[0:4]
    return 100.0 * (t1 - t0) / (r1 - r0) if r1 > r0 else None


def main():
    nb_path = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    ty = np.where(k >= 1.0, "C", "P")
    chain = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4, optiontypes_ttms=(ty,) * 4)
    p = sv.LOGSV_BTC_PARAMS
    sets = [sv.LogSvParams(sigma0=p.sigma0 + 1e-3 * j, theta=p.theta, kappa1=p.kappa1 + 1e-2 * j, kappa2=p.kappa2, beta=p.beta,
                           volvol=p.volvol - 1e-2 * j) for j in range(8)]

    def timed(fn):
        fn()
        fn()
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(ts))

    out = {"nb_path": nb_path, "steps": None, "calls": calls}
    routes = (("frozen", False),) if os.environ.get("SVMC_BENCH_FROZEN_ONLY") == "1" else (("frozen", False), ("hbm", True))
    for tag, in_hbm in routes:
        res = sv.draw_fixed_randoms_on_device(ttms, nb_path=nb_path, nb_steps_per_year=360, seed=10, in_hbm=in_hbm)
        out["steps"] = int(sum(res.nb_steps))

        def one(res=res):
            q = sets[0]
            return sv.logsv_mc_chain_pricer_fixed_randoms(W0s=res, W1s=None, dts=None, v0=q.sigma0, theta=q.theta, kappa1=q.kappa1,
                                                          kappa2=q.kappa2, beta=q.beta, volvol=q.volvol, vol_backbone_etas=np.ones(4),
                                                          return_ivols=True, **chain)
        row = {"one_set_with_ivols_ms": timed(one)}
        for n_sets in (6, 7, 8):
            row[f"{n_sets}_sets_with_ivols_ms"] = timed(lambda: sv.logsv_mc_chain_pricer_fixed_randoms_batch(
                params_list=sets[:n_sets], W0s=res, return_ivols=True, **chain))
        if not in_hbm:
            got = one()
            want = sv.logsv_mc_chain_pricer(v0=sets[0].sigma0, theta=sets[0].theta, kappa1=sets[0].kappa1, kappa2=sets[0].kappa2,
                                            beta=sets[0].beta, volvol=sets[0].volvol, vol_backbone_etas=np.ones(4), nb_path=nb_path,
                                            nb_steps_per_year=360, seed=10, **chain)
            row["bit_equal_to_logsv_mc_chain_pricer"] = bool(all(np.array_equal(a, b) for a, b in zip(got[0] + got[1], want[0] + want[1])))
            row["hbm_bytes_held_for_randoms"] = 0
            from stochvolmodels_amd import _lib
            _lib.check(_lib.load().svmc_clock_probe_arm(1))
            from stochvolmodels_amd.engine import option_type_codes
            q0 = sets[0]
            for _ in range(30):            # un-replayed launches (a captured launch never carries the probe)
                res.price_logsv_chain(ttms, chain["forwards"], chain["discfactors"], [k] * 4, [option_type_codes(ty)] * 4, q0.sigma0,
                                      q0.theta, q0.kappa1, q0.kappa2, q0.beta, q0.volvol, np.ones(4), True, 1, use_graph=False)
            row["clock_mhz_in_kernel_one_set"] = clock_mhz_of_last_launch()
            _lib.check(_lib.load().svmc_clock_probe_arm(0))
        else:
            row["hbm_bytes_held_for_randoms"] = int(16 * nb_path * out["steps"])
        out[tag] = row
        res.free()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
