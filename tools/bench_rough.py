#!/usr/bin/env python
"""
Rough-LogSV kernel timing on one MI355X: 2^20 paths x 360 steps, N = 1, 2, 3 factors, with the
normals drawn in the kernel and streamed from HBM.  One JSON object per line.

    python tools/bench_rough.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from stochvolmodels_amd.engine import get_engine  # noqa: E402

RULES = {1: ([1e-3], [1.0]), 2: ([0.00181, 1.524], [0.9, 1.7]), 3: ([0.0772, 5.19, 108.46], [0.777, 1.554, 8.516])}


def main():
    n, nb = 1 << 20, 360
    eng = get_engine(n)
    h = 1.0 / nb
    for nf, (nodes, weights) in RULES.items():
        nodes, weights = np.array(nodes), np.array(weights)
        v0 = np.full(nf, 0.377 / weights.sum())
        args = (nb, h, nodes, weights, v0, 0.347, 1.29, 1.93, 2.45 / np.hypot(2.45, 1.81), float(np.hypot(2.45, 1.81)))
        z0, z1 = eng.fill_normals(nb, 11)
        for mode, kw in (("device_rng", dict(seed=5)), ("streamed", dict(z0_ptr=z0, z1_ptr=z1))):
            for _ in range(2):
                eng.rough_logsv(*args, **kw)
            eng.start_kernel_timing()
            for _ in range(5):
                eng.rough_logsv(*args, **kw)
            ms = np.mean(eng.stop_kernel_timing()["rough_logsv_kernel"])
            x, _, y = eng.get_state()
            print(json.dumps(dict(kernel="rough_logsv_kernel", n_factors=nf, mode=mode, paths=n, steps=nb, ms=float(ms),
                                  path_steps_per_s=n * nb / (ms * 1e-3), mean_spot=float(np.exp(x).mean()),
                                  mean_qvar=float(y.mean()))))


def chain():
    """a whole rough chain through the pricer (the rough calibration engine's objective evaluation): 4 expiries, resident
    randoms, every expiry re-simulated from time 0 -- all expiries in one launch (svmc_rough_logsv_chain) against one
    launch per expiry"""
    import time

    import stochvolmodels_amd as sv
    from stochvolmodels_amd.pricers import logsv_pricer as lp
    ttms = np.array([1 / 12, 0.25, 0.5, 1.0])
    k = np.linspace(0.7, 1.3, 13)
    chain_kw = dict(ttms=ttms, forwards=np.ones(4), discfactors=np.ones(4), strikes_ttms=(k,) * 4,
                    optiontypes_ttms=(np.where(k >= 1.0, "C", "P"),) * 4)
    p = sv.LogSvParams(sigma0=0.8, theta=1.0, kappa1=3.0, kappa2=3.0, beta=0.15, volvol=1.8, H=0.1)
    p.approximate_kernel(T=1.0)
    pars = dict(sigma0=p.sigma0, theta=p.theta, kappa1=p.kappa1, kappa2=p.kappa2, beta=p.beta, orthog_vol=p.volvol,
                weights=p.weights, nodes=p.nodes)
    for n in (4000, 100_000):
        Z0, Z1, grids = sv.get_randoms_for_rough_vol_chain_valuation(ttms, nb_path=n, nb_steps_per_year=360, seed=10)
        res = sv.upload_rough_randoms(Z0, Z1)
        out = {}
        for flag in (False, True, False, True):
            lp.ROUGH_CHAIN_ONE_LAUNCH = flag
            f = lambda: sv.rough_logsv_mc_chain_pricer_fixed_randoms(Z0=res, Z1=None, timegrids=grids, **chain_kw, **pars)   # noqa: E731
            for _ in range(5):
                f()
            ts = []
            for _ in range(100):
                t0 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t0)
            out.setdefault(flag, []).append(1e3 * float(np.median(ts)))
        lp.ROUGH_CHAIN_ONE_LAUNCH = True
        print(json.dumps(dict(config="rough LogSV chain, 4 expiries x 13 strikes, 3 factors, resident randoms", paths=n,
                              steps=[int(g.size) - 1 for g in grids], one_launch_ms=out[True], launch_per_expiry_ms=out[False])))
        res.free()


if __name__ == "__main__":
    if "--chain" in sys.argv:
        chain()
    else:
        main()
