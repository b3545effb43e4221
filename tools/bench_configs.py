#!/usr/bin/env python
"""
Secondary measurements on one MI355X (not the driver's bench line): BASELINE configs C1, C3, a single-GPU share
of C4, plus a breakdown of the C2 call.  Prints one JSON object per line.

    python tools/bench_configs.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import stochvolmodels_amd as sv  # noqa: E402
from stochvolmodels_amd.engine import get_engine  # noqa: E402


def timed(fn, reps=5, warm=2):
    """median wall time of `reps` synchronised calls (the median, not the mean: a process's first few hundred calls
    contain one ~60 ms stall of the ROCm runtime, profiles/r02_runtime_stall.txt, which a mean over 3-10 calls would
    spread over whatever leg it happens to land in)"""
    for _ in range(warm):
        fn()
    eng_sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        eng_sync()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def eng_sync():
    import ctypes as C
    from stochvolmodels_amd import _lib
    _lib.load().svmc_stream_synchronize(None)


def main():
    kk = np.linspace(0.5, 1.5, 21)
    types = np.where(kk >= 1.0, "C", "P")
    res = []
    # C1: Heston Euler 10k x 100
    k5, t5 = np.array([0.8, 0.9, 1.0, 1.1, 1.2]), np.array(["P", "P", "C", "C", "C"])
    dt, _ = timed(lambda: sv.heston_mc_chain_pricer(ttms=np.array([1.0]), forwards=np.ones(1), discfactors=np.ones(1),
                                                    strikes_ttms=(k5,), optiontypes_ttms=(t5,), v0=0.04, theta=0.04,
                                                    kappa=4.0, rho=-0.5, volvol=0.4, nb_path=10_000,
                                                    nb_steps_per_year=99, seed=1), reps=20)
    res.append(dict(config="C1 Heston Euler 10k x 100, 5 strikes", ms=1e3 * dt, path_steps_per_s=1e6 / dt))
    # C3: Heston QE / Euler 2^22 x 512 (4 x 128), 4 x 21 strikes, two parameter sets
    ttms = np.array([0.25, 0.5, 0.75, 1.0])
    for tag, p in (("base", sv.HestonParams()), ("btc", sv.BTC_HESTON_PARAMS)):
        for scheme in ("qe", "euler"):
            n = 1 << 22
            dt, out = timed(lambda: sv.heston_mc_chain_pricer(
                ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(kk,) * 4,
                optiontypes_ttms=(types,) * 4, v0=p.v0, theta=p.theta, kappa=p.kappa, rho=p.rho, volvol=p.volvol,
                nb_path=n, scheme=scheme, nb_steps_per_year=508, seed=3), reps=3, warm=1)
            res.append(dict(config=f"C3 Heston {scheme} {tag} 2^22 x 512, 4x21 strikes", ms=1e3 * dt,
                            path_steps_per_s=n * 512 / dt, atm_price_1y=float(out[0][3][10])))
    # C4 share: LogSV 2^21 paths (one rank of 8) x 1024 steps, 8 expiries x 21 strikes
    P = sv.LOGSV_BTC_PARAMS
    ttms8 = np.arange(1, 9) / 8.0
    fw = 67000.0 * np.exp(0.05 * ttms8)
    strikes8 = tuple(f * np.linspace(0.6, 1.6, 21) for f in fw)
    types8 = tuple(np.where(k >= f, "C", "P") for k, f in zip(strikes8, fw))
    chain8 = sv.OptionChain(ttms=ttms8, forwards=fw, strikes_ttms=strikes8, optiontypes_ttms=types8, ids=None)
    n = 1 << 21
    dt, _ = timed(lambda: sv.LogSVPricer().model_mc_price_chain(chain8, P, nb_path=n, nb_steps=1016, seed=4), reps=3, warm=1)
    res.append(dict(config="C4 one-rank share: LogSV 2^21 x 1024 (8 x 128), 8x21 strikes", ms=1e3 * dt,
                    path_steps_per_s=n * 1024 / dt))
    # C2 breakdown
    n = 1 << 20
    chain = sv.OptionChain.slice_to_chain(ttm=1.0, forward=1.0, strikes=kk, optiontypes=types)
    pricer = sv.LogSVPricer()
    dt_all, _ = timed(lambda: pricer.model_mc_price_chain(chain, P, nb_path=n, nb_steps=1023, seed=5), reps=10)
    eng = get_engine(n)
    dt_step, _ = timed(lambda: eng.logsv_rng(1024, 1 / 1024, P.theta, P.kappa1, P.kappa2, P.beta, P.volvol, 1.0, True, 5, 0, 0), reps=10)
    dt_pay, _ = timed(lambda: sv.compute_mc_vars_payoff(x0=np.zeros(8), sigma0=np.ones(8), qvar0=np.zeros(8), ttm=1.0,
                                                        forward=1.0, strikes_ttm=kk, optiontypes_ttm=types), reps=10)
    res.append(dict(config="C2 breakdown", whole_call_ms=1e3 * dt_all, stepping_kernel_ms=1e3 * dt_step,
                    overhead_ms=1e3 * (dt_all - dt_step), tiny_payoff_call_ms=1e3 * dt_pay))
    # fixed-randoms through the host API (PCIe-inclusive)
    nb, npth = 256, 1 << 18
    W0s, W1s, dts = sv.get_randoms_for_chain_valuation(np.array([0.25]), nb_path=npth, nb_steps_per_year=1020, seed=1)
    c1 = sv.OptionChain.slice_to_chain(ttm=0.25, forward=1.0, strikes=kk, optiontypes=types)
    dt, _ = timed(lambda: sv.logsv_mc_chain_pricer_fixed_randoms(
        ttms=c1.ttms, forwards=c1.forwards, discfactors=c1.discfactors, strikes_ttms=c1.strikes_ttms,
        optiontypes_ttms=c1.optiontypes_ttms, W0s=W0s, W1s=W1s, dts=dts, v0=P.sigma0, theta=P.theta, kappa1=P.kappa1,
        kappa2=P.kappa2, beta=P.beta, volvol=P.volvol, vol_backbone_etas=np.ones(1)), reps=3, warm=1)
    steps = W0s[0].shape[0]
    res.append(dict(config=f"fixed randoms from host (pageable numpy), 2^18 x {steps}", ms=1e3 * dt,
                    path_steps_per_s=npth * steps / dt, host_to_device_GBps=16.0 * npth * steps / dt / 1e9))
    res_dev = sv.upload_fixed_randoms(W0s, W1s, dts)
    dt, _ = timed(lambda: sv.logsv_mc_chain_pricer_fixed_randoms(
        ttms=c1.ttms, forwards=c1.forwards, discfactors=c1.discfactors, strikes_ttms=c1.strikes_ttms,
        optiontypes_ttms=c1.optiontypes_ttms, W0s=res_dev, W1s=None, dts=None, v0=P.sigma0, theta=P.theta,
        kappa1=P.kappa1, kappa2=P.kappa2, beta=P.beta, volvol=P.volvol, vol_backbone_etas=np.ones(1)), reps=10, warm=2)
    res.append(dict(config=f"fixed randoms RESIDENT in HBM (upload_fixed_randoms), 2^18 x {steps}", ms=1e3 * dt,
                    path_steps_per_s=npth * steps / dt, hbm_GBps=16.0 * npth * steps / dt / 1e9))
    # C5 analytic side
    g5 = sv.OptionChain(ttms=np.array([0.125, 0.25, 0.375, 0.5]), forwards=np.ones(4), strikes_ttms=(kk,) * 4,
                        optiontypes_ttms=(types,) * 4, ids=None)
    dt, _ = timed(lambda: pricer.price_chain(g5, P), reps=5, warm=1)
    res.append(dict(config="C5 analytic side: LogSV affine-expansion chain 4 x 21 strikes, 1000-point phi grid", ms=1e3 * dt,
                    prices_per_s=84 / dt))
    # C5 analytic side, the five parameter sets of the sweep: one by one and in one batch of launches
    g = np.load(os.path.join(ROOT, "tests", "golden", "analytic.npz"))
    sets = []
    for tag in ("btc", "readme", "quick", "test", "fig3"):
        v = [float(a) for a in g[f"logsv_{tag}_params"]]
        sets.append(sv.LogSvParams(sigma0=v[0], theta=v[1], kappa1=v[2], kappa2=v[3], beta=v[4], volvol=v[5]))
    c5 = sv.OptionChain(ttms=np.array([0.25, 0.5, 0.75, 1.0]), forwards=np.ones(4), strikes_ttms=(kk,) * 4,
                        optiontypes_ttms=(types,) * 4, ids=None)
    dt1, _ = timed(lambda: [pricer.price_chain(c5, p_) for p_ in sets], reps=5, warm=1)
    dtb, _ = timed(lambda: pricer.price_chain_batch(c5, sets), reps=5, warm=1)
    res.append(dict(config="C5 analytic side, 5 parameter sets x (4 x 21 strikes)", one_by_one_ms=1e3 * dt1, batched_ms=1e3 * dtb))
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
